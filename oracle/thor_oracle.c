/* thor_oracle.c - TEST INFRASTRUCTURE (never linked into, imported or called by the product).
 *
 * Plain-C restatement of the reference's kernel-level arithmetic on the per-block encode path,
 * used as the checker for the kernel-level entry points of libthor_hip.so (tests/ and
 * __graft_entry__.smoke()).  Every function cites the reference code it follows.  The restatement
 * is pinned: tests/test_oracle_c.py checks it against known-answer vectors recorded from the real
 * reference functions (tests/golden/kat.npz and kat2.npz, produced by tests/golden/gen_kat.py / gen_kat2.py
 * through oracle/_ref/libthorref.so; both the scalar C and the SIMD variant of every reference function are run
 * by the generators and must agree), and, when /root/reference is present, against the live reference.
 * The frame-level oracle is the reference encoder itself (oracle/_ref/Thorenc).
 *
 * 8-bit samples (and the 16-bit twins of the SAD / interpolation / transform-unit functions), 4:2:0.  Build: gcc -O2 -std=c99 -fPIC -shared -ffp-contract=off
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int ilog2(unsigned v) { int n = 0; while (v >>= 1) n++; return n; }
static int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* ---- SAD: sad_calc, enc/encode_block.c:417-428 ----------------------------------------- */
unsigned orc_sad(const uint8_t* a, int astride, const uint8_t* b, int bstride, int w, int h) {
  unsigned s = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) s += (unsigned)abs((int)a[y * astride + x] - (int)b[y * bstride + x]);
  return s;
}

/* ---- widesad_calc, enc/encode_block.c:430-453 (scalar: first strictly smaller SAD over the offsets -3 -1 0 1 3) and widesad_calc_simd,
 * enc/enc_kernels.c:84-113 (16x16: min over (sad << 3 | code), code rising with the offset) - both keep the LEFTMOST offset among equal SADs */
unsigned orc_widesad(const uint8_t* a, int astride, const uint8_t* b, int bstride, int w, int h, int* x) {
  static const int off[5] = {-3, -1, 0, 1, 3};
  unsigned best = 1u << 31;
  int bx = 0;
  for (int k = 0; k < 5; k++) {
    const unsigned s = orc_sad(a, astride, b + off[k], bstride, w, h);
    if (s < best) { best = s; bx = off[k]; }
  }
  *x = bx;
  return best;
}

/* ---- ME bit estimate: quote_mv_bits, enc/encode_block.c:467-515 ------------------------ */
static int mvlen(int d) {
  d = abs(d);
  if (d < 1) return 2;
  if (d < 2) return 4;
  if (d < 4) return 5;
  if (d < 36) return 6 + ((d - 4) >> 3);
  return 11 + ((d - 36) >> 4);
}
int orc_quote_mv_bits(int dy, int dx) { return mvlen(dx) + mvlen(dy); }

/* ---- quarter-pel luma prediction: get_inter_prediction_luma, common/inter_prediction.c:117-181,
 *      taps common/common_kernels.c:1905-1917.  ref points at the block's co-located sample. ---- */
static const int TAPS_STD[4][6] = {{0, 0, 64, 0, 0, 0}, {1, -7, 55, 19, -5, 1}, {1, -7, 38, 38, -7, 1}, {1, -5, 19, 55, -7, 1}};
static const int TAPS_BI[4][6] = {{0, 0, 64, 0, 0, 0}, {2, -10, 59, 17, -5, 1}, {1, -8, 39, 39, -8, 1}, {1, -5, 17, 59, -10, 2}};
void orc_interp_luma(uint8_t* dst, int dstride, const uint8_t* ref, int rstride, int w, int h, int mvx, int mvy, int sign,
                     int bipred, int pic_w, int pic_h, int xpos, int ypos) {
  if (sign) { mvx = -mvx; mvy = -mvy; }
  int fy = mvy & 3, fx = mvx & 3, iy = mvy >> 2, ix = mvx >> 2;
  if (iy > pic_h - ypos) iy = pic_h - ypos;
  if (iy < -xpos - h) iy = -xpos - h; /* sic: xpos, inter_prediction.c:129 */
  if (ix > pic_w - xpos) ix = pic_w - xpos;
  if (ix < -xpos - w) ix = -xpos - w;
  const int(*T)[6] = bipred ? TAPS_BI : TAPS_STD;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const uint8_t* p = ref + (y + iy) * rstride + x + ix;
      int v;
      if (!fx && !fy) v = p[0];
      else if (fx == 2 && fy == 2 && bipred < 2) {
        int s = p[-rstride] + p[-rstride + 1] + p[-1] + 2 * p[0] + 2 * p[1] + p[2] + p[rstride - 1] + 2 * p[rstride] +
                2 * p[rstride + 1] + p[rstride + 2] + p[2 * rstride] + p[2 * rstride + 1];
        v = clampi((s + 8) >> 4, 0, 255);
      } else {
        int acc = 0;
        for (int n = 0; n < 6; n++) {
          int col = 0;
          for (int m = 0; m < 6; m++) col += T[fy][m] * p[(m - 2) * rstride + n - 2];
          acc += T[fx][n] * col;
        }
        v = clampi((acc + 2048) >> 12, 0, 255);
      }
      dst[y * dstride + x] = (uint8_t)v;
    }
}

/* ---- HEVC-style integer DCT bases (common/transform.c:37-241) generated from the cosine table --- */
static int hevc_cos(int k) {
  static const int mag[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                              61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
  k &= 127;
  if (k <= 32) return mag[k];
  if (k <= 64) return -mag[64 - k];
  if (k <= 96) return -mag[k - 64];
  return mag[128 - k];
}
static int dctm(int n, int i, int j) { return hevc_cos(i * (2 * j + 1) * (32 / n)); }

/* forward transform: transform(), common/transform.c:245-309.  block: size x size residual,
 * coeff: compact q x q (q = min(size,16)). */
void orc_fwd_transform(const int16_t* block, int16_t* coeff, int size, int fast, int bitdepth) {
  int q = size < 16 ? size : 16, size1 = size, scale = 1;
  int16_t* in = (int16_t*)malloc(sizeof(int16_t) * 32 * 32);
  int16_t* tmp = (int16_t*)malloc(sizeof(int16_t) * 16 * 32);
  if (size > (32 >> fast)) {
    size1 = 32 >> fast;
    scale = size / size1;
    for (int i = 0; i < size1; i++)
      for (int j = 0; j < size1; j++) {
        int16_t s = 0;
        for (int m = 0; m < scale; m++)
          /* what the reference executes is transform_simd (common/common_kernels.c:1482-1562): saturating accumulation with `fast`
           * (:1526-1531), a plain wrapping int16_t sum without (:1545-1550); the scalar transform.c:262-277 saturates always */
          for (int n = 0; n < scale; n++) {
            const int v = s + block[(i * scale + m) * size + j * scale + n];
            s = fast ? (int16_t)clampi(v, -16384, 16383) : (int16_t)v;
          }
        in[i * size1 + j] = s;
      }
  } else
    memcpy(in, block, sizeof(int16_t) * size * size);
  int sh1 = ilog2(size) + ilog2(scale) + bitdepth - 8, sh2 = ilog2(size1) + 5;
  for (int i = 0; i < q; i++)
    for (int j = 0; j < size1; j++) {
      int s = 0;
      for (int k = 0; k < size1; k++) s += dctm(size1, i, k) * in[j * size1 + k];
      tmp[i * size1 + j] = (int16_t)((s + (1 << (sh1 - 1))) >> sh1);
    }
  for (int i = 0; i < q; i++)
    for (int j = 0; j < q; j++) {
      int s = 0;
      for (int k = 0; k < size1; k++) s += dctm(size1, i, k) * tmp[j * size1 + k];
      coeff[i * q + j] = (int16_t)((s + (1 << (sh2 - 1))) >> sh2);
    }
  free(in);
  free(tmp);
}

/* zigzag scan position of (i,j) in an n x n block (common/common_tables.c:29-66) */
static void zigzag(int n, int* zz) {
  int idx = 0;
  for (int d = 0; d <= 2 * (n - 1); d++)
    for (int t = 0; t < n; t++) {
      int i = (d & 1) ? t : n - 1 - t, j = d - i;
      if (j < 0 || j >= n) continue;
      zz[i * n + j] = idx++;
    }
}

static const int QSCALE[6] = {26214, 23302, 20560, 18396, 16384, 14564};
static const int DQSCALE[6] = {40, 45, 51, 57, 64, 72};

/* ---- the sub-block tests of the early-skip check, enc/encode_block.c:2123-2229 ----------------------------------------------------------------
 * luma (check_early_skip_sub_block): residual, 2x2 average (size > 4), forward transform of size / 2, any |coefficient| > (int)(0.5 * thr * first quantiser
 * level) - thr is a float, the product is formed in double as the reference's C does. */
int orc_early_skip_sub(const uint8_t* org, int ostride, const uint8_t* pred, int pstride, int size, int qp, float thr, int bitdepth) {
  int16_t res[32 * 32], tmp[16 * 16], coef[16 * 16];
  for (int i = 0; i < size; i++)
    for (int j = 0; j < size; j++) res[i * size + j] = (int16_t)((int)org[i * ostride + j] - (int)pred[i * pstride + j]);
  int n = size;
  double rel = thr;
  const int16_t* in = res;
  if (size > 4) {
    n = size / 2;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++)
        tmp[i * n + j] = (int16_t)((res[2 * i * size + 2 * j] + res[2 * i * size + 2 * j + 1] + res[(2 * i + 1) * size + 2 * j] + res[(2 * i + 1) * size + 2 * j + 1] + 2) >> 2);
    in = tmp;
    rel = 0.5 * thr;
  }
  orc_fwd_transform(in, coef, n, 0, bitdepth);
  const int shift2 = 21 - ilog2((unsigned)n) + qp / 6;
  const int threshold = (int)(rel * ((double)(1 << shift2) / (double)QSCALE[qp % 6]));
  int flag = 0;
  for (int k = 0; k < n * n; k++) flag |= abs((int)coef[k]) > threshold;
  return flag;
}
/* chroma (check_early_skip_sub_blockC -> calc_cbp_simd, enc/enc_kernels.c:828-907: the form the encoder EXECUTES, use_simd = 1): int16 column sums of the
 * residual with 16-bit wrap-around; 16 / 8 wide: any |sum| > threshold as int16; 4 wide: per column PAIR (2k, 2k+1) the test is sum[2k+1] + |sum[2k]| >
 * threshold (arithmetic on sign-extended halves) - NOT the scalar calc_cbp's |sum[2k] + sum[2k+1]|. */
int orc_early_skip_subC(const uint8_t* org, int ostride, const uint8_t* pred, int pstride, int size, int qp, float thr, int bitdepth) {
  const int shift2 = 21 - 5 + qp / 6;
  const int threshold = ((int)(thr * ((double)(1 << shift2) / (double)QSCALE[qp % 6]))) << (bitdepth - 8);
  int16_t col[16];
  for (int j = 0; j < size; j++) {
    int16_t s = 0;
    for (int i = 0; i < size; i++) s = (int16_t)(s + (int16_t)((int)org[i * ostride + j] - (int)pred[i * pstride + j]));
    col[j] = s;
  }
  if (size == 4) {
    for (int k = 0; k < 2; k++) {
      const int lo = col[2 * k], hi = col[2 * k + 1];
      if (hi + (int)(int16_t)(lo < 0 ? -lo : lo) > threshold) return 1;
    }
    return 0;
  }
  for (int j = 0; j < size; j++) {
    const int16_t a = (int16_t)(col[j] < 0 ? -col[j] : col[j]);
    if (a > (int16_t)threshold) return 1;
  }
  return 0;
}

/* quantize(), enc/encode_block.c:84-160 (no weight matrix).  coeff/coeffq compact q x q. */
int orc_quantize(const int16_t* coeff, int16_t* coeffq, int qp, int size, int intra_block) {
  int q = size < 16 ? size : 16, N = q * q, zz[256], sc[256], sq[256];
  zigzag(q, zz);
  int64_t scale = QSCALE[qp % 6];
  int shift2 = 21 - ilog2(size) + qp / 6;
  for (int i = 0; i < N; i++) { sc[zz[i]] = coeff[i]; sq[i] = 0; }
  int64_t off = (intra_block ? 38 : -26) * ((int64_t)1 << (shift2 - 8));
  int level = 0, pos = N - 1;
  while (level == 0 && pos >= 0) {
    int64_t l = (int64_t)abs(sc[pos]) * scale + off;
    level = (int)((l > 0 ? l : -l) >> shift2);
    pos--;
  }
  int last = level ? pos + 1 : pos, cbp = 0, level_mode = 1;
  int64_t o0 = (int64_t)(intra_block ? 102 : 51) << (shift2 - 8), o1 = (int64_t)(intra_block ? 115 : 90) << (shift2 - 8);
  for (pos = 0; pos <= last; pos++) {
    int c = sc[pos];
    int64_t ac = scale * abs(c);
    int l0 = (int)(ac >> shift2);
    int lev = (int)((ac + ((l0 > (1 - level_mode)) ? o1 : o0)) >> shift2);
    sq[pos] = c < 0 ? -lev : lev;
    cbp |= lev != 0;
    if (level_mode) { if (lev == 0) level_mode = 0; } else if (lev > 1) level_mode = 1;
  }
  for (int i = 0; i < N; i++) coeffq[i] = (int16_t)sq[zz[i]];
  return cbp;
}

/* dequantize(), common/common_block.c:45-73 (compact in, compact out) */
void orc_dequantize(const int16_t* coeffq, int16_t* rcoeff, int qp, int size) {
  int q = size < 16 ? size : 16, ls = qp / 6, rs = ilog2(size) - 1;
  int64_t scale = DQSCALE[qp % 6];
  for (int i = 0; i < q * q; i++) {
    int64_t c = coeffq[i];
    rcoeff[i] = ls >= rs ? (int16_t)((c * scale) * ((int64_t)1 << (ls - rs))) : (int16_t)((c * scale + ((int64_t)1 << (rs - ls - 1))) >> (rs - ls));
  }
}

/* inverse_transform(), common/transform.c:411-494: rcoeff compact q x q -> block size x size */
void orc_inv_transform(const int16_t* rcoeff, int16_t* block, int size, int bitdepth) {
  int n = size < 32 ? size : 32, scale = size / n, q = n < 16 ? n : 16, sh2 = 20 - bitdepth;
  int16_t* tmp = (int16_t*)malloc(sizeof(int16_t) * 16 * 32);
  for (int i = 0; i < q; i++)
    for (int j = 0; j < n; j++) {
      int s = 0;
      for (int k = 0; k < q; k++) s += dctm(n, k, j) * rcoeff[k * q + i];
      tmp[i * n + j] = (int16_t)clampi((s + 64) >> 7, -32768, 32767);
    }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      int s = 0;
      for (int k = 0; k < q; k++) s += dctm(n, k, j) * tmp[k * n + i];
      int r = clampi((s + (1 << (sh2 - 1))) >> sh2, -32768, 32767);
      for (int m = 0; m < scale; m++)
        for (int x = 0; x < scale; x++) block[(scale * i + m) * size + scale * j + x] = (int16_t)r;
    }
  free(tmp);
}

/* One transform unit end to end (encode_and_reconstruct_block_inter non-split branch,
 * enc/encode_block.c:1319-1330): residual, T, Q, IQ, IT, reconstruct_block (common_block.c:75-84).
 * coeff_type bit1 = intra-frame flag. Returns cbp; coefq compact. */
int orc_code_tu(const uint8_t* org, const uint8_t* pred, int size, int qp, int coeff_type, int fast, int16_t* coefq, uint8_t* rec) {
  int n2 = size * size;
  int16_t* res = (int16_t*)malloc(sizeof(int16_t) * n2);
  int16_t* rb = (int16_t*)malloc(sizeof(int16_t) * n2);
  int16_t co[256], rc[256];
  for (int i = 0; i < n2; i++) res[i] = (int16_t)((int)org[i] - (int)pred[i]);
  orc_fwd_transform(res, co, size, fast, 8);
  int cbp = orc_quantize(co, coefq, qp, size, (coeff_type >> 1) & 1);
  if (cbp) {
    orc_dequantize(coefq, rc, qp, size);
    orc_inv_transform(rc, rb, size, 8);
    for (int i = 0; i < n2; i++) rec[i] = (uint8_t)clampi(rb[i] + pred[i], 0, 255);
  } else
    memcpy(rec, pred, n2);
  free(res);
  free(rb);
  return cbp;
}

/* ---- 16-bit samples (the reference's _hbd instances: SAMPLE = uint16_t, the same source compiled a second time,
 *      enc/encode_block_hbd.c / common/inter_prediction_hbd.c ...).  Pinned by tests/golden/kat4.npz (bitdepth 10). ---- */
unsigned orc_sad16(const uint16_t* a, int astride, const uint16_t* b, int bstride, int w, int h) {
  unsigned s = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) s += (unsigned)abs((int)a[y * astride + x] - (int)b[y * bstride + x]);
  return s;
}
/* widesad_calc_simd_hbd (enc/enc_kernels.c:84-113 compiled with HBD): 16-bit samples, same rule */
unsigned orc_widesad16(const uint16_t* a, int astride, const uint16_t* b, int bstride, int w, int h, int* x) {
  static const int off[5] = {-3, -1, 0, 1, 3};
  unsigned best = 1u << 31;
  int bx = 0;
  for (int k = 0; k < 5; k++) {
    const unsigned s = orc_sad16(a, astride, b + off[k], bstride, w, h);
    if (s < best) { best = s; bx = off[k]; }
  }
  *x = bx;
  return best;
}
void orc_interp_luma16(uint16_t* dst, int dstride, const uint16_t* ref, int rstride, int w, int h, int mvx, int mvy, int sign,
                       int bipred, int pic_w, int pic_h, int xpos, int ypos, int bitdepth) {
  const int maxv = (1 << bitdepth) - 1;
  if (sign) { mvx = -mvx; mvy = -mvy; }
  int fy = mvy & 3, fx = mvx & 3, iy = mvy >> 2, ix = mvx >> 2;
  if (iy > pic_h - ypos) iy = pic_h - ypos;
  if (iy < -xpos - h) iy = -xpos - h; /* sic: xpos, inter_prediction.c:129 */
  if (ix > pic_w - xpos) ix = pic_w - xpos;
  if (ix < -xpos - w) ix = -xpos - w;
  const int(*T)[6] = bipred ? TAPS_BI : TAPS_STD;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const uint16_t* p = ref + (y + iy) * rstride + x + ix;
      int v;
      if (!fx && !fy) v = p[0];
      else if (fx == 2 && fy == 2 && bipred < 2) {
        int s = p[-rstride] + p[-rstride + 1] + p[-1] + 2 * p[0] + 2 * p[1] + p[2] + p[rstride - 1] + 2 * p[rstride] +
                2 * p[rstride + 1] + p[rstride + 2] + p[2 * rstride] + p[2 * rstride + 1];
        v = clampi((s + 8) >> 4, 0, maxv);
      } else {
        int acc = 0;
        for (int n = 0; n < 6; n++) {
          int col = 0;
          for (int m = 0; m < 6; m++) col += T[fy][m] * p[(m - 2) * rstride + n - 2];
          acc += T[fx][n] * col;
        }
        v = clampi((acc + 2048) >> 12, 0, maxv);
      }
      dst[y * dstride + x] = (uint16_t)v;
    }
}
int orc_code_tu16(const uint16_t* org, const uint16_t* pred, int size, int qp, int coeff_type, int fast, int16_t* coefq, uint16_t* rec,
                  int bitdepth) {
  int n2 = size * size;
  const int maxv = (1 << bitdepth) - 1;
  int16_t* res = (int16_t*)malloc(sizeof(int16_t) * n2);
  int16_t* rb = (int16_t*)malloc(sizeof(int16_t) * n2);
  int16_t co[256], rc[256];
  for (int i = 0; i < n2; i++) res[i] = (int16_t)((int)org[i] - (int)pred[i]);
  orc_fwd_transform(res, co, size, fast, bitdepth);
  int cbp = orc_quantize(co, coefq, qp, size, (coeff_type >> 1) & 1);
  if (cbp) {
    orc_dequantize(coefq, rc, qp, size);
    orc_inv_transform(rc, rb, size, bitdepth);
    for (int i = 0; i < n2; i++) rec[i] = (uint16_t)clampi(rb[i] + pred[i], 0, maxv);
  } else
    memcpy(rec, pred, sizeof(uint16_t) * n2);
  free(res);
  free(rb);
  return cbp;
}

/* ---- VLC lengths (enc/putvlc.c:73-160) and write_coeff bit count (enc/write_bits.c:145-241) ---- */
static int vlc_len(int n, unsigned cn) {
  if (n == 6 || n == 7) {
    if (cn == 0) return 2;
    if (n == 6) { cn++; n = 2; }
    else {
      if (cn == 1) return 3;
      if (cn < 4) return 4;
      cn += 4; n = 3;
    }
  }
  if (n <= 5) {
    unsigned t = 1u << n;
    if (cn < 5 * t) return 1 + n + (int)(cn >> n);
    return (5 - n) + 1 + 2 * ilog2(cn - 5 * t + t);
  }
  if (n == 8) return cn < 6 ? 2 + (int)(cn >> 1) : 5;
  if (n == 10) return 1 + 2 * ilog2(cn + 1);
  return cn == (unsigned)(n - 10) ? n - 10 : (int)cn + 1;
}
int orc_coeff_bits(const int16_t* coeff, int size, int type) {
  int q = size < 16 ? size : 16, N = q * q, zz[256], s[256], bits = 0;
  zigzag(q, zz);
  for (int i = 0; i < N; i++) s[zz[i]] = coeff[i];
  int chroma = type & 1, intra = (type >> 1) & 1, adaptive = intra && !chroma;
  unsigned eob = chroma ? 0 : 2;
  int runtab = (chroma && size <= 8) ? 10 : 6;
  int last = N - 1;
  while (last > 0 && !s[last]) last--;
  int pos = 0;
  if (chroma) {
    if (last == 0 && abs(s[0]) == 1) return 2;
    bits += 1;
  }
  int level_mode = 1, level = 1, c;
  while (pos <= last) {
    if (level_mode)
      while (pos <= last && level > 0) {
        c = s[pos++];
        level = abs(c);
        bits += vlc_len(adaptive, (unsigned)level) + (level > 0);
        if (!chroma) adaptive = level > 3;
      }
    int run = 0;
    c = 0;
    while (c == 0 && pos <= last) {
      c = s[pos++];
      run += !c;
      if (c) {
        level = abs(c);
        unsigned cn = level == 1 ? (unsigned)(run * 5 / 4) : (unsigned)(run * 5 + 4);
        bits += vlc_len(runtab, cn + (cn >= eob));
        level_mode = level > 1;
        bits += level > 1 ? vlc_len(0, (unsigned)((level - 2) * 2 + (c < 0))) : 1;
        run = 0;
      }
    }
  }
  if (pos < N && level_mode) { bits += vlc_len(adaptive, 0); pos++; }
  if (pos < N) bits += vlc_len(runtab, eob);
  return bits;
}

/* ---- encoder_speed > 0 bilinear sub-pel approximations -------------------------------------------------
 * sad_calc_fasthalf (enc/encode_block.c:174-283; SIMD twin enc_kernels.c:330-513): SADs of the eight half-pel
 * neighbours of the block at b, each neighbour built from rounding ("avg", (x+y+1)>>1) and truncating ("rd",
 * (x+y)>>1) two-sample averages; returns the smallest SAD and writes its offset (+-2 quarter-pel units).
 * Evaluation order of the strict-'<' scan: top (initial), down, right, left, tl, tr, br, bl. */
static int av2(int x, int y) { return (x + y + 1) >> 1; }
static int rd2(int x, int y) { return (x + y) >> 1; }
unsigned orc_fast_halfpel(const uint8_t* a, const uint8_t* b, int as, int bs, int w, int h, int* bx, int* by) {
  unsigned s[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* top, down, right, left, tl, tr, br, bl */
  for (int i = 0; i < h; i++)
    for (int j = 0; j < w; j++) {
      const uint8_t* c = b + i * bs + j;
#define B(dy, dx) ((int)c[(dy) * bs + (dx)])
      int o = a[i * as + j];
      int hl = av2(B(0, -1), B(0, 0)), hr = av2(B(0, 0), B(0, 1));
      int vu = av2(B(-2, 0), B(1, 0)), vd = av2(B(-1, 0), B(2, 0));
      int wl = av2(B(0, -2), B(0, 1)), wr = av2(B(0, -1), B(0, 2));
      int ptl = rd2(rd2(rd2(av2(B(-2, -1), B(1, -1)), vu), rd2(av2(B(-1, -2), B(-1, 1)), wl)), rd2(av2(B(-1, -1), B(-1, 0)), hl));
      int ptr = rd2(rd2(rd2(vu, av2(B(-2, 1), B(1, 1))), rd2(wr, av2(B(-1, -1), B(-1, 2)))), rd2(av2(B(-1, 0), B(-1, 1)), hr));
      int pbl = rd2(rd2(rd2(vd, av2(B(-1, -1), B(2, -1))), rd2(wl, av2(B(1, -2), B(1, 1)))), rd2(av2(B(1, -1), B(1, 0)), hl));
      int pbr = rd2(rd2(rd2(vd, av2(B(-1, 1), B(2, 1))), rd2(wr, av2(B(1, -1), B(1, 2)))), rd2(hr, av2(B(1, 0), B(1, 1))));
      s[0] += abs(o - av2(B(0, 0), B(-1, 0))); s[1] += abs(o - av2(B(0, 0), B(1, 0)));
      s[2] += abs(o - hr); s[3] += abs(o - hl);
      s[4] += abs(o - ptl); s[5] += abs(o - ptr); s[6] += abs(o - pbr); s[7] += abs(o - pbl);
#undef B
    }
  static const int ox[8] = {0, 0, 2, -2, -2, 2, 2, -2}, oy[8] = {-2, 2, 0, 0, -2, -2, 2, 2};
  unsigned best = s[0];
  int k = 0;
  for (int q = 1; q < 8; q++)
    if (s[q] < best) { best = s[q]; k = q; }
  *bx = ox[k]; *by = oy[k];
  return best;
}

/* sad_calc_fastquarter (enc/encode_block.c:286-415): the eight quarter-pel neighbours of a position whose half-pel
 * phase is (*x != 0, *y != 0) on entry; order of the strict-'<' scan: top (initial), tl, tr, left, right, bl, down, br. */
unsigned orc_fast_quarterpel(const uint8_t* o_, const uint8_t* r_, int os, int rs, int w, int h, int* x, int* y) {
  unsigned s[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* top, tl, tr, left, right, bl, down, br */
  const int hx = *x != 0, hy = *y != 0;
  for (int i = 0; i < h; i++)
    for (int j = 0; j < w; j++) {
      const uint8_t* c = r_ + i * rs + j;
      int o = o_[i * os + j], a = c[0], d = c[1], f = c[rs];
      int p[8];
      if (hx && hy) {
        int e = c[rs + 1], ad = av2(a, d), de = av2(d, e), af = av2(a, f), fe = av2(f, e);
        p[1] = (ad + af) >> 1; p[0] = (de + a) >> 1; p[2] = (ad + de) >> 1; p[3] = (ad + f) >> 1; p[4] = (ad + e) >> 1;
        p[5] = (af + fe) >> 1; p[6] = (de + f) >> 1; p[7] = (de + fe) >> 1;
      } else if (hx) {
        int b = c[-rs], cc = c[-rs + 1], e = c[rs + 1], ad = av2(a, d), de = av2(d, e), dc = av2(d, cc), af = av2(a, f), ab = av2(a, b);
        p[1] = (ad + ab) >> 1; p[0] = (dc + a) >> 1; p[2] = (ad + dc) >> 1; p[3] = (ad + a) >> 1; p[4] = (ad + d) >> 1;
        p[5] = (ad + af) >> 1; p[6] = (af + d) >> 1; p[7] = (ad + de) >> 1;
      } else if (hy) {
        int e = c[rs + 1], g = c[rs - 1], hh = c[-1], ad = av2(a, d), af = av2(a, f), fe = av2(f, e), ah = av2(a, hh), gf = av2(g, f);
        p[1] = (ah + af) >> 1; p[0] = (af + a) >> 1; p[2] = (ad + af) >> 1; p[3] = (gf + a) >> 1; p[4] = (ad + f) >> 1;
        p[5] = (af + gf) >> 1; p[6] = (af + f) >> 1; p[7] = (af + fe) >> 1;
      } else {
        int b = c[-rs], hh = c[-1], ad = av2(a, d), af = av2(a, f), ah = av2(a, hh), ab = av2(a, b);
        p[1] = (ah + ab) >> 1; p[0] = (ab + a) >> 1; p[2] = (ad + ab) >> 1; p[3] = (ah + a) >> 1; p[4] = (ad + a) >> 1;
        p[5] = (ah + af) >> 1; p[6] = (af + a) >> 1; p[7] = (af + ad) >> 1;
      }
      for (int q = 0; q < 8; q++) s[q] += abs(o - p[q]);
    }
  static const int ox[8] = {0, -1, 1, -1, 1, -1, 0, 1}, oy[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
  unsigned best = s[0];
  int k = 0;
  for (int q = 1; q < 8; q++)
    if (s[q] < best) { best = s[q]; k = q; }
  *x = ox[k]; *y = oy[k];
  return best;
}

/* ---- CLPF -------------------------------------------------------------------------------------------------
 * constrain (common/common_block.c:217-221) and clpf_sample (:315-321): cross-shaped 8-tap low-pass of the
 * constrained differences, weights 1 3 1 3 | 3 1 3 1 for A B C D | E F G H, rounded towards zero on a 1/16 grid. */
static int orc_constrain(int diff, int thr, int damping) {
  if (!thr) return 0;
  int ad = abs(diff), lim = thr - (ad >> (damping - ilog2((unsigned)thr)));
  if (lim < 0) lim = 0;
  int m = ad < lim ? ad : lim;
  return diff < 0 ? -m : m;
}
int orc_clpf_sample(int X, int A, int B, int C, int D, int E, int F, int G, int H, int s, int dmp) {
  int delta = orc_constrain(A - X, s, dmp) + 3 * orc_constrain(B - X, s, dmp) + orc_constrain(C - X, s, dmp) + 3 * orc_constrain(D - X, s, dmp) +
              3 * orc_constrain(E - X, s, dmp) + orc_constrain(F - X, s, dmp) + 3 * orc_constrain(G - X, s, dmp) + orc_constrain(H - X, s, dmp);
  return (8 + delta - (delta < 0)) >> 4;
}
/* detect_multi_clpf (enc/encode_block.c:2584-2621): squared error of one size x size block against the original,
 * unfiltered and filtered with strengths 1, 2, 4 (<< shift), neighbours clamped to the frame. sum[0..3] += ... */
void orc_detect_multi_clpf(const uint8_t* rec, const uint8_t* org, int x0, int y0, int width, int height, int ostride, int rstride,
                           int* sum, int shift, int size, int dmp) {
  unsigned s[4] = {0, 0, 0, 0};
  for (int y = y0; y < y0 + size; y++)
    for (int x = x0; x < x0 + size; x++) {
      int ym2 = y - 2 < 0 ? 0 : y - 2, ym1 = y - 1 < 0 ? 0 : y - 1, yp1 = y + 1 > height - 1 ? height - 1 : y + 1, yp2 = y + 2 > height - 1 ? height - 1 : y + 2;
      int xm2 = x - 2 < 0 ? 0 : x - 2, xm1 = x - 1 < 0 ? 0 : x - 1, xp1 = x + 1 > width - 1 ? width - 1 : x + 1, xp2 = x + 2 > width - 1 ? width - 1 : x + 2;
      int O = org[y * ostride + x], X = rec[y * rstride + x];
      int A = rec[ym2 * rstride + x], B = rec[ym1 * rstride + x], Cc = rec[y * rstride + xm2], D = rec[y * rstride + xm1];
      int E = rec[y * rstride + xp1], F = rec[y * rstride + xp2], G = rec[yp1 * rstride + x], H = rec[yp2 * rstride + x];
      s[0] += (unsigned)((O - X) * (O - X));
      for (int k = 1; k < 4; k++) {
        int Y = X + orc_clpf_sample(X, A, B, Cc, D, E, F, G, H, (1 << (k - 1)) << shift, dmp);
        s[k] += (unsigned)((O - Y) * (O - Y));
      }
    }
  for (int k = 0; k < 4; k++) sum[k] += (int)(s[k] >> (shift * 2));
}
