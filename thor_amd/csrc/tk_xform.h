// tk_xform.h - residual, forward transform, quantise, dequantise, inverse transform, reconstruct.
// Scalar specifications followed: common/transform.c:245-309 (transform, incl. the box-sum
// down-scaling of 64/128 blocks and the int16 stage-1 truncation), :411-494 (inverse, 32x32 kernel
// + replication for 64/128), enc/encode_block.c:84-160 (quantize), common/common_block.c:45-73
// (dequantize), :75-84 (reconstruct_block).  Coefficient blocks are kept COMPACT (stride
// qsize = min(size,16)) instead of the reference's stride-`size` layout.
#pragma once
#include "tk_common.h"

#ifndef TK_PROF_T0
#if defined(THOR_PROF) && !TK_HOST && !defined(THOR_PROF_NOMACROS)
#ifdef THOR_PROF_WALL
#define TK_CYC() ((long long)wall_clock64())
#else
#define TK_CYC() ((long long)__builtin_readcyclecounter())
#endif
#define TK_PROF_T0() long long pt0_ = TK_CYC()
#define TK_PROF_ADD(ws, id) do { if (t.rank == 0) (ws)->prof[id] += TK_CYC() - pt0_; } while (0)
#define TK_PROF_MARK(v) long long v = TK_CYC()
#define TK_PROF_ACC(ws, id, v) do { if (t.rank == 0) (ws)->prof[id] += TK_CYC() - (v); } while (0)
#define TK_PROF_CNT(ws, id) do { if (t.rank == 0) (ws)->prof[id] += 1; } while (0)
#else
#define TK_PROF_T0() do {} while (0)
#define TK_PROF_ADD(ws, id) do {} while (0)
#define TK_PROF_MARK(v) do {} while (0)
#define TK_PROF_ACC(ws, id, v) do {} while (0)
#define TK_PROF_CNT(ws, id) do {} while (0)
#endif
#endif

namespace tk {

// Constant tables of a workgroup (LDS on the GPU), shared by all of its wavefronts: the 32-point DCT basis (the
// N-point basis is its rows 0, 32/N, 2*32/N, ... restricted to the first N columns - HEVC nesting property) and the
// scan tables 4x4 | 8x8 | 16x16 (Team::izz points at `izz`).
struct XformTabs {
  alignas(16) int16_t dct32[1024];
  int16_t izz[336];
  // transposed N-point bases for the inverse transform, MT_N[j][k] = M_N[k][j]: 32 | 16 | 8 | 4 -point at offsets 0, 1024, 1280, 1344
  alignas(16) int16_t dctT[1360];
};
// The transform stages take the basis from Team::izz (which points at XformTabs::izz of the workgroup's tables): no pointer
// load from the workspace per stage.  Host simulation: Team::izz is unused, the tables come from ws->tabs.
static_assert(offsetof(XformTabs, izz) == 1024 * sizeof(int16_t), "dct32 must directly precede izz");
static_assert(offsetof(XformTabs, dctT) == (1024 + 336) * sizeof(int16_t), "dctT must directly follow izz");
#if TK_HOST
#define TK_DCT32(t, ws) ((ws)->tabs->dct32)
#define TK_DCTT(t, ws) ((ws)->tabs->dctT)
#else
#define TK_DCT32(t, ws) (TK_LDS_PTR((t).izz) - 1024)
#define TK_DCTT(t, ws) (TK_LDS_PTR((t).izz) + 336)
#endif
TK_DEV int dctT_off(int n) { return n == 32 ? 0 : n == 16 ? 1024 : n == 8 ? 1280 : 1344; }
struct XformWs {
  // `in`: (down-scaled) residual fed to the core transform, row-major in[row*size1 + col].  It is dead after forward
  // stage 1, so the inverse transform's stage-1 buffer (itmp, TRANSPOSED [sample j][coef col i], 32*16 entries) lives in
  // the same storage.  Every stage is a set of dot products of two CONTIGUOUS int16 vectors (a basis row and a data row),
  // read with 16-byte ds_read and multiplied two terms at a time (v_dot2_i32_i16).
  // (the scalars come first: the arrays end the structure, so that the motion search's window - which borrows in | tmp | coef while
  // no transform is running - continues into the bytes that follow the structure, SmallWs::win_extra)
  int flag;               // team-shared scalar result
  long long* prof;        // cycle counters (THOR_PROF builds)
  const XformTabs* tabs;  // workgroup-shared constant tables
  alignas(16) int16_t in[32 * 32];
  alignas(16) int16_t tmp[16 * 32];   // forward stage-1 output, TRANSPOSED: tmp[coef i][row j] (stride size1)
  alignas(16) int16_t coef[16 * 16];  // forward coefficients, compact; reused for the de-quantised ones (rcoef, TRANSPOSED)
};
static_assert(sizeof(XformWs) == offsetof(XformWs, coef) + 16 * 16 * sizeof(int16_t), "XformWs must end with its arrays");

// entry (i, q) of the N-point basis, log2(32/N) = rs
TK_DEV int dct_at(const XformWs* ws, int rs, int i, int q) { return ws->tabs->dct32[((i << rs) << 5) + q]; }
// Fill the shared tables: thread `rank` of `size` cooperating threads (the whole workgroup on the GPU; the caller
// synchronises afterwards).  Once per workgroup - the tables never change.
TK_DEV void xform_tables_fill(XformTabs* tb, int rank, int size) {
  for (int k = rank; k < 1024; k += size) tb->dct32[k] = TK_TAB.dct32[k];
  for (int k = rank; k < 336; k += size) tb->izz[k] = k < 16 ? TK_TAB.izz4[k] : (k < 80 ? TK_TAB.izz8[k - 16] : TK_TAB.izz16[k - 80]);
  for (int k = rank; k < 1360; k += size) {
    const int n = k < 1024 ? 32 : k < 1280 ? 16 : k < 1344 ? 8 : 4, rs = 5 - ilog2((unsigned)n);
    const int e = k - dctT_off(n), j = e / n, q = e - j * n;
    tb->dctT[k] = TK_TAB.dct32[((q << rs) << 5) + j];   // M_n[q][j]
  }
}

// sum_{q<n} a[q] * b[q] of two contiguous int16 vectors in the team's transform workspace / tables (n = 4, 8, 16 or 32; both
// 2n-byte aligned up to 16).  rot: chunk the lane starts with (spreads the lanes of a wave over the LDS banks).
#if !TK_HOST
typedef short __attribute__((ext_vector_type(2))) tk_s16x2;
#endif
TK_DEV int dot_i16(const lds_i16* a, const lds_i16* b, int n, int rot) {
#if TK_HOST
  (void)rot;
  int s = 0;
  for (int q = 0; q < n; q++) s += (int)a[q] * (int)b[q];
  return s;
#else
  typedef uint32_t __attribute__((ext_vector_type(4))) v4;
  typedef uint32_t __attribute__((ext_vector_type(2))) v2;
  auto d2 = [](uint32_t x, uint32_t y, int c) -> int {
    tk_s16x2 xv, yv;
    __builtin_memcpy(&xv, &x, 4); __builtin_memcpy(&yv, &y, 4);
    return __builtin_amdgcn_sdot2(xv, yv, c, false);
  };
  if (n == 4) {
    const v2 x = *(const TK_LDS v2*)a, y = *(const TK_LDS v2*)b;
    return d2(x.y, y.y, d2(x.x, y.x, 0));
  }
  const int nch = n >> 3;
  int s = 0;
  for (int c = 0; c < nch; c++) {
    const int cc = ((c + rot) & (nch - 1)) << 3;
    const v4 x = *(const TK_LDS v4*)(a + cc), y = *(const TK_LDS v4*)(b + cc);
    s = d2(x.w, y.w, d2(x.z, y.z, d2(x.y, y.y, d2(x.x, y.x, s))));
  }
  return s;
#endif
}

TK_DEV void fwd_core(const Team t, XformWs* ws, int size1, int qsize, int shift_1);

// NS consecutive samples (NS * sizeof(PIX) = 2, 4, 8 or 16 bytes, aligned to that) as ONE memory instruction.  SP: address space of p.
template <int SP, typename PIX, int NS> TK_DEV void load_samples(const PIX* p, int* out) {
  enum { B = NS * (int)sizeof(PIX) };
  static_assert(B == 2 || B == 4 || B == 8 || B == 16, "piece size");
  PIX v[NS];
#if TK_HOST
  __builtin_memcpy(v, p, B);
#else
  if constexpr (B == 2) { const uint16_t x = *(typename SpT<SP, const uint16_t>::ptr)spc<SP>(p); __builtin_memcpy(v, &x, 2); }
  else {
    typedef uint32_t __attribute__((ext_vector_type(B / 4))) vec_t;
    const vec_t x = *(typename SpT<SP, const vec_t>::ptr)spc<SP>(p);
    __builtin_memcpy(v, &x, B);
  }
#endif
  for (int q = 0; q < NS; q++) out[q] = (int)v[q];
}
template <int SP, typename PIX, int NS> TK_DEV void store_samples(PIX* p, const int* in) {
  enum { B = NS * (int)sizeof(PIX) };
  PIX v[NS];
  for (int q = 0; q < NS; q++) v[q] = (PIX)in[q];
#if TK_HOST
  __builtin_memcpy(p, v, B);
#else
  if constexpr (B == 2) { uint16_t x; __builtin_memcpy(&x, v, 2); *(typename SpT<SP, uint16_t>::ptr)spc<SP>(p) = x; }
  else {
    typedef uint32_t __attribute__((ext_vector_type(B / 4))) vec_t;
    vec_t x;
    __builtin_memcpy(&x, v, B);
    *(typename SpT<SP, vec_t>::ptr)spc<SP>(p) = x;
  }
#endif
}
// box sum of one scale x scale cell of the residual, rows of NS = scale samples per memory instruction.  What the reference EXECUTES is
// transform_simd (common/common_kernels.c:1482-1562, use_simd = 1): with `fast` (encoder_speed > 1) the running sum saturates to
// [-16384, 16383] after every sample, in raster order (:1526-1531); WITHOUT it the cell is a plain int16_t accumulation that wraps (:1545-1550) -
// the scalar code (transform.c:262-277) saturates in both cases.  The two differ only when a cell's sum leaves +-16383: never up to 10 bits
// (16 x 1023), but at 12 bits a 4x4 cell of a 128x128 block reaches 16 x 4095 (found by the 12-bit known answers of round 6).
template <int SP, typename PIX, int NS> TK_DEV int box_residual(const PIX* org, int ostride, const PIX* pred, int pstride, int sat) {
  int sum = 0;
#if !TK_HOST
#pragma nounroll   // a rolled row loop: unrolled, the compiler fetches and unpacks all NS x NS samples at once (VGPR budget of the 8-bit kernel)
#endif
  for (int m = 0; m < NS; m++) {
    int o[NS], p[NS];
    load_samples<SP, PIX, NS>(org + m * ostride, o);
    load_samples<SP, PIX, NS>(pred + m * pstride, p);
    for (int n = 0; n < NS; n++) { const int v = (int16_t)sum + (int16_t)(o[n] - p[n]); sum = sat ? clampi(v, -16384, 16383) : (int)(int16_t)v; }
  }
  return sum;
}
template <int SP, typename PIX, int NS> TK_DEV void box_recon(int r, const PIX* pred, int pstride, PIX* rec, int rstride, int bitdepth) {
#if !TK_HOST
#pragma nounroll
#endif
  for (int m = 0; m < NS; m++) {
    int p[NS];
    load_samples<SP, PIX, NS>(pred + m * pstride, p);
    for (int n = 0; n < NS; n++) p[n] = sat_pix(r + p[n], bitdepth);
    store_samples<SP, PIX, NS>(rec + m * rstride, p);
  }
}

// Forward transform of (org - pred) -> ws->coef (qsize x qsize compact).
template <typename PIX, int SP>
TK_DEV void fwd_transform(const Team t, XformWs* ws, const PIX* org_, int ostride, const PIX* pred_, int pstride,
                          int size, int fast, int bitdepth) {
  const auto org = spc<SP>(org_);
  const auto pred = spc<SP>(pred_);
  const int qsize = size < kMaxQuant ? size : kMaxQuant;
  int size1 = size, scale = 1;
  if (size > (32 >> fast)) {
    size1 = 32 >> fast;
    scale = size / size1;
  }
  // residual (+ optional box sum, common_kernels.c:1521-1556)
  lds_i16* const in_l = TK_LDS_PTR(ws->in);
#ifndef TK_NOVEC
  {
    // four residuals (scale 1) or one box of scale x scale samples per lane and step, rows in single memory instructions; every TU /
    // block position and stride is a multiple of four samples - checked (wave-uniform), sample by sample otherwise
    const int S = (int)sizeof(PIX);
    const unsigned al = (unsigned)(uintptr_t)org_ | (unsigned)(uintptr_t)pred_ | (unsigned)(ostride * S) | (unsigned)(pstride * S);
    const int step = scale == 1 ? 4 : scale;
    if (tk_uniform(!(al & (unsigned)(step * S - 1)))) {
      if (scale == 1) {
        const int ppr = size1 >> 2, lg = ilog2((unsigned)ppr);
        for (int k = t.rank; k < ppr * size1; k += t.size) {
          const int i = k >> lg, j = (k & (ppr - 1)) << 2;
          int o[4], p[4];
          load_samples<SP, PIX, 4>(org_ + i * ostride + j, o);
          load_samples<SP, PIX, 4>(pred_ + i * pstride + j, p);
          int16_t r4[4];
          for (int q = 0; q < 4; q++) r4[q] = (int16_t)(o[q] - p[q]);
#if TK_HOST
          __builtin_memcpy(ws->in + i * size1 + j, r4, 8);
#else
          typedef uint32_t __attribute__((ext_vector_type(2))) v2;
          v2 x;
          __builtin_memcpy(&x, r4, 8);
          *(TK_LDS v2*)(in_l + i * size1 + j) = x;
#endif
        }
      } else {
        for (int k = t.rank; k < size1 * size1; k += t.size) {
          int i, j;
          split2(mk_pow2(size1), k, i, j);
          const PIX* o = org_ + i * scale * ostride + j * scale;
          const PIX* p = pred_ + i * scale * pstride + j * scale;
          in_l[i * size1 + j] = (int16_t)(scale == 2 ? box_residual<SP, PIX, 2>(o, ostride, p, pstride, fast)
                                          : scale == 4 ? box_residual<SP, PIX, 4>(o, ostride, p, pstride, fast) : box_residual<SP, PIX, 8>(o, ostride, p, pstride, fast));
        }
      }
      t.sync();
      fwd_core(t, ws, size1, qsize, ilog2(size) + ilog2(scale) + bitdepth - 8);
      return;
    }
  }
#endif
  for (int k = t.rank; k < size1 * size1; k += t.size) {
    int i, j;
    split2(mk_pow2(size1), k, i, j);
    int sum = 0;
    if (scale == 1) {
      sum = (int16_t)((int)org[i * ostride + j] - (int)pred[i * pstride + j]);
    } else {
      for (int m = 0; m < scale; m++)
        for (int n = 0; n < scale; n++) {
          int y = i * scale + m, x = j * scale + n;
          int r = (int16_t)((int)org[y * ostride + x] - (int)pred[y * pstride + x]);
          sum = fast ? clampi((int16_t)sum + r, -16384, 16383) : (int)(int16_t)((int16_t)sum + r);   // transform_simd: see box_residual
        }
    }
    in_l[i * size1 + j] = (int16_t)sum;
  }
  t.sync();
  fwd_core(t, ws, size1, qsize, ilog2(size) + ilog2(scale) + bitdepth - 8);
}

// Core 2-D transform of the size1 x size1 block in ws->in (row-major) -> ws->coef (qsize x qsize).
// Stage 1 (rows):   T[j][i] = (sum_q M[i][q] * in[j][q] + add1) >> shift1   -> stored transposed tmp[i][j] (stride size1)
// Stage 2 (cols):   coef[i][j] = (sum_q M[i][q] * T[q][j] + add2) >> shift2 = dot(M row i, tmp row j)
TK_DEV void fwd_core(const Team t, XformWs* ws, int size1, int qsize, int shift_1) {
  const int rs = 5 - ilog2((unsigned)size1);
  const int add_1 = 1 << (shift_1 - 1);
  const int shift_2 = ilog2(size1) + 5;
  const int add_2 = 1 << (shift_2 - 1);
  const lds_i16* const in = TK_LDS_PTR(ws->in);
  const lds_i16* const dct = TK_DCT32(t, ws);
  lds_i16* const tmp = TK_LDS_PTR(ws->tmp);
  lds_i16* const coef = TK_LDS_PTR(ws->coef);
  const Pow2 d1 = mk_pow2(size1), dq = mk_pow2(qsize);
  for (int k = t.rank; k < qsize * size1; k += t.size) {
    int i, j;
    split2(d1, k, i, j);  // row j fastest
    const int sum = dot_i16(dct + ((i << rs) << 5), in + j * size1, size1, j);
    tmp[i * size1 + j] = (int16_t)((sum + add_1) >> shift_1);
  }
  t.sync();
  for (int k = t.rank; k < qsize * qsize; k += t.size) {
    int i, j;
    split2(dq, k, i, j);
    const int sum = dot_i16(dct + ((i << rs) << 5), tmp + j * size1, size1, j);
    coef[i * qsize + j] = (int16_t)((sum + add_2) >> shift_2);
  }
  t.sync();
}

// Same core transform on an explicit int16 block already stored (row-major) in ws->in (early skip).
TK_DEV void fwd_transform_block(const Team t, XformWs* ws, int size, int bitdepth) {
  fwd_core(t, ws, size, size, ilog2(size) + bitdepth - 8);
}

// quantize (encode_block.c:84-160): zigzag scan with the `level_mode` state machine, evaluated W = team.size scan
// positions at a time.
// The level_mode state is a 2-state automaton whose per-coefficient transition is one of
// {identity, ->0, ->1} (never a swap, because the level under mode 1 is >= the level under mode 0),
// so the state entering a position is the target of the nearest earlier constant transition; a
// ballot + count-leading-zeros finds it.  With W = 1 this is literally the reference's serial loop.
template <int SC>
TK_DEV int quantize_team(const Team t, XformWs* ws, int16_t* coefq_, int qp, int size, int intra_block) {
  const auto coefq = spc<SC>(coefq_);
  const int qsize = size < kMaxQuant ? size : kMaxQuant;
  const int N = qsize * qsize;
  const IzzRef izzr = izz_ref(t, qsize);
  const int scale = quant_scale(qp % 6);
  const int shift2 = 21 - ilog2(size) + qp / 6;
  const int offl = intra_block ? (38 << (shift2 - 8)) : -(26 << (shift2 - 8));
  const int off0 = (intra_block ? 102 : 51) << (shift2 - 8);
  const int off1 = (intra_block ? 115 : 90) << (shift2 - 8);
  const int W = t.size;
  const lds_i16* const coef = TK_LDS_PTR(ws->coef);
  // last_pos: highest position whose level under the "last position" offset is non-zero
  int last_pos = -1;
  for (int base = 0; base < N; base += W) {
    const int p = base + t.rank;
    int nz = 0;
    if (p < N) {
      int l = iabs((int)coef[izzr.z[p]]) * scale + offl;
      nz = ((l > 0 ? l : -l) >> shift2) != 0;
    }
    const unsigned long long m = team_ballot(t, nz);
    if (m) last_pos = base + top_set(m);
  }
  int carry = 1, cbp = 0;
  for (int base = 0; base < N; base += W) {
    const int p = base + t.rank;
    const int active = p < N && p <= last_pos;
    int c = 0, lev0 = 0, lev1 = 0;
    if (active) {
      c = coef[izzr.z[p]];
      const int ac = scale * iabs(c);
      const int level0 = ac >> shift2;
      lev1 = (ac + (level0 > 0 ? off1 : off0)) >> shift2;   // level_mode == 1
      lev0 = (ac + (level0 > 1 ? off1 : off0)) >> shift2;   // level_mode == 0
    }
    const int to0 = active && lev1 == 0, to1 = active && lev0 > 1;
    const unsigned long long mK = team_ballot(t, to0 || to1), mV = team_ballot(t, to1);
    const int j = prev_set(mK, t.rank);
    const int mode = j < 0 ? carry : (int)((mV >> j) & 1ull);
    const int lev = mode ? lev1 : lev0;
    if (p < N) coefq[izzr.z[p]] = (int16_t)(c < 0 ? -lev : lev);
    cbp |= team_ballot(t, active && lev != 0) != 0ull;
    const int jj = top_set(mK);
    if (jj >= 0) carry = (int)((mV >> jj) & 1ull);
  }
  t.sync();
  return cbp;
}

// dequantize (common_block.c:45-73): coefq -> ws->rcoef, int16 truncation as in the reference.
template <int SC>
TK_DEV void dequantize(const Team t, XformWs* ws, const int16_t* coefq_, int qp, int size) {
  const auto coefq = spc<SC>(coefq_);
  const int qsize = size < kMaxQuant ? size : kMaxQuant;
  const int lshift = qp / 6, rshift = ilog2(size) - 1;
  const int64_t scale = dequant_scale(qp % 6);
  lds_i16* const rcoef = TK_LDS_PTR(ws->coef);
  const int lgq = ilog2((unsigned)qsize);
  for (int k = t.rank; k < qsize * qsize; k += t.size) {
    int64_t c = coefq[k];
    int16_t r;
    if (lshift >= rshift) r = (int16_t)((c * scale) << (lshift - rshift));
    else r = (int16_t)((c * scale + ((int64_t)1 << (rshift - lshift - 1))) >> (rshift - lshift));
    rcoef[((k & (qsize - 1)) << lgq) + (k >> lgq)] = r;  // TRANSPOSED (rcoef[col][row]); aliases coef: every lane has read its coefq[k] from the
                                                          // caller's buffer, not from here
  }
  t.sync();
}

// inverse transform of ws->rcoef + prediction -> rec (saturated), replicating for 64/128.
template <typename PIX, int SP>
TK_DEV void inv_transform_recon(const Team t, XformWs* ws, const PIX* pred_, int pstride, PIX* rec_, int rstride,
                                int size, int bitdepth) {
  const auto pred = spc<SP>(pred_);
  const auto rec = spc<SP>(rec_);
  const int n = size < 32 ? size : 32;
  const int scale = size / n;
  const int qsize = n < kMaxQuant ? n : kMaxQuant;
  const int rs = 5 - ilog2((unsigned)n);
  lds_i16* const itmp = TK_LDS_PTR(ws->in);  // aliases `in`
  const lds_i16* const mt = TK_DCTT(t, ws) + dctT_off(n);   // MT[j][k] = M[k][j], row pitch n
  const lds_i16* const rcoef = TK_LDS_PTR(ws->coef);        // transposed by dequantize: rcoef[col i][row k]
  const int shift_2 = 20 - bitdepth, add_2 = 1 << (shift_2 - 1);
  const Pow2 dn = mk_pow2(n);
  (void)rs;
  // stage 1: itmp[j][i] = clip((sum_k M[k][j]*rcoef[k][i] + 64) >> 7)   i < qsize, j < n   (stored [sample j][coef col i])
  for (int k = t.rank; k < qsize * n; k += t.size) {
    int i, j;
    split2(dn, k, i, j);
    const int sum = dot_i16(mt + j * n, rcoef + i * qsize, qsize, j);
    itmp[j * qsize + i] = (int16_t)clampi((sum + 64) >> 7, -32768, 32767);
  }
  t.sync();
#ifndef TK_NOVEC
  {
    const int S = (int)sizeof(PIX);
    const unsigned al = (unsigned)(uintptr_t)pred_ | (unsigned)(uintptr_t)rec_ | (unsigned)(pstride * S) | (unsigned)(rstride * S);
    const int step = scale == 1 ? 4 : scale;
    if (tk_uniform(!(al & (unsigned)(step * S - 1)))) {
      if (scale == 1) {
        // four neighbouring samples of a row per lane and step: the stage-1 row is shared, prediction and reconstruction move in one
        // memory instruction each
        const int ppr = n >> 2, lg = ilog2((unsigned)ppr);
        for (int k = t.rank; k < ppr * n; k += t.size) {
          const int i = k >> lg, j = (k & (ppr - 1)) << 2;
          int p[4];
          load_samples<SP, PIX, 4>(pred_ + i * pstride + j, p);
#if !TK_HOST
#pragma unroll
#endif
          for (int q = 0; q < 4; q++) {
            const int sum = dot_i16(mt + (j + q) * n, itmp + i * qsize, qsize, j + q);
            p[q] = sat_pix(clampi((sum + add_2) >> shift_2, -32768, 32767) + p[q], bitdepth);
          }
          store_samples<SP, PIX, 4>(rec_ + i * rstride + j, p);
        }
      } else {
        for (int k = t.rank; k < n * n; k += t.size) {
          int i, j;
          split2(dn, k, i, j);
          const int sum = dot_i16(mt + j * n, itmp + i * qsize, qsize, j);
          const int r = clampi((sum + add_2) >> shift_2, -32768, 32767);
          const PIX* p = pred_ + scale * i * pstride + scale * j;
          PIX* o = rec_ + scale * i * rstride + scale * j;
          if (scale == 2) box_recon<SP, PIX, 2>(r, p, pstride, o, rstride, bitdepth);
          else if (scale == 4) box_recon<SP, PIX, 4>(r, p, pstride, o, rstride, bitdepth);
          else box_recon<SP, PIX, 8>(r, p, pstride, o, rstride, bitdepth);
        }
      }
      t.sync();
      return;
    }
  }
#endif
  for (int k = t.rank; k < n * n; k += t.size) {
    int i, j;
    split2(dn, k, i, j);
    const int sum = dot_i16(mt + j * n, itmp + i * qsize, qsize, j);
    int r = clampi((sum + add_2) >> shift_2, -32768, 32767);
    for (int m = 0; m < scale; m++)
      for (int x = 0; x < scale; x++) {
        int yy = scale * i + m, xx = scale * j + x;
        rec[yy * rstride + xx] = (PIX)sat_pix(r + (int)pred[yy * pstride + xx], bitdepth);
      }
  }
  t.sync();
}

// Row-wise copy with V-byte pieces (V = 16, 8 or 4): one piece per lane and step instead of one sample.
template <int SD, int SS, int V> TK_DEV void copy_rows_vec(const Team t, void* dst_, int dpitch, const void* src_, int spitch, int rowbytes, int h) {
  const int ppr = rowbytes / V;   // pieces per row (a power of two for every block this is used on, or any number: divided below)
#if TK_HOST
  for (int k = t.rank; k < ppr * h; k += t.size) {
    const int i = k / ppr, j = k - i * ppr;
    __builtin_memcpy((char*)dst_ + (size_t)i * dpitch + j * V, (const char*)src_ + (size_t)i * spitch + j * V, V);
  }
#else
  typedef uint32_t __attribute__((ext_vector_type(V / 4))) vec_t;
  const auto dst = spc<SD>((char*)dst_);
  const auto src = spc<SS>((const char*)src_);
  const int lg = (ppr & (ppr - 1)) ? -1 : ilog2((unsigned)ppr);
  for (int k = t.rank; k < ppr * h; k += t.size) {
    int i, j;
    if (lg >= 0) { i = k >> lg; j = k & (ppr - 1); } else { i = k / ppr; j = k - i * ppr; }
    *(typename SpT<SD, vec_t>::ptr)(dst + i * dpitch + j * V) = *(typename SpT<SS, const vec_t>::ptr)(src + i * spitch + j * V);
  }
#endif
}
template <int SD, int SS, typename PIX>
TK_DEV void copy_block(const Team t, PIX* dst_, int dstride, const PIX* src_, int sstride, int w, int h) {
  dst_ = tk_uniform_ptr(dst_); src_ = tk_uniform_ptr(src_); dstride = tk_uniform(dstride); sstride = tk_uniform(sstride); w = tk_uniform(w); h = tk_uniform(h);
  // widest piece that divides the row and keeps every access aligned (wave-uniform): a 64x64 block moves in 4 steps of 16-byte
  // pieces instead of 64 steps of single samples
  const int S = (int)sizeof(PIX);
#ifdef TK_NOVEC   // A/B build: sample by sample
  const unsigned al = 1u;
#else
  const unsigned al = (unsigned)(uintptr_t)dst_ | (unsigned)(uintptr_t)src_ | (unsigned)(dstride * S) | (unsigned)(sstride * S) | (unsigned)(w * S);
#endif
  if (!(al & 15u)) { copy_rows_vec<SD, SS, 16>(t, dst_, dstride * S, src_, sstride * S, w * S, h); return; }
  if (!(al & 7u)) { copy_rows_vec<SD, SS, 8>(t, dst_, dstride * S, src_, sstride * S, w * S, h); return; }
  if (!(al & 3u)) { copy_rows_vec<SD, SS, 4>(t, dst_, dstride * S, src_, sstride * S, w * S, h); return; }
  const auto dst = spc<SD>(dst_);
  const auto src = spc<SS>(src_);
  if ((w & (w - 1)) == 0) {
    const Pow2 pw = mk_pow2(w);
    for (int k = t.rank; k < w * h; k += t.size) {
      int i, j;
      split2(pw, k, i, j);
      dst[i * dstride + j] = src[i * sstride + j];
    }
  } else {  // frame-edge rectangles (skip blocks)
    for (int k = t.rank; k < w * h; k += t.size) {
      int i = k / w, j = k - i * w;
      dst[i * dstride + j] = src[i * sstride + j];
    }
  }
}

// One transform unit: residual -> T -> Q -> (IQ -> IT -> recon | recon = pred). Returns cbp bit.
// coeff_type: bit0 chroma, bit1 = (frame_type == I) [sic: frame type, Appendix B.6].
// SP: address space of org / pred / rec (all three belong to the same coding block), SC: of coefq.
template <typename PIX, int SP, int SC>
TK_DEVNI int code_tu_sp(const Team t, XformWs* ws, const PIX* org, int ostride, const PIX* pred, int pstride, PIX* rec,
                   int rstride, int size, int qp, int coeff_type, int fast, int16_t* coefq, int bitdepth) {
  org = tk_uniform_ptr(org); pred = tk_uniform_ptr(pred); rec = tk_uniform_ptr(rec); coefq = tk_uniform_ptr(coefq); ws = tk_uniform_ptr(ws);
  ostride = tk_uniform(ostride); pstride = tk_uniform(pstride); rstride = tk_uniform(rstride); size = tk_uniform(size);
  qp = tk_uniform(qp); coeff_type = tk_uniform(coeff_type); fast = tk_uniform(fast); bitdepth = tk_uniform(bitdepth);
  TK_PROF_T0();
  fwd_transform<PIX, SP>(t, ws, org, ostride, pred, pstride, size, fast, bitdepth);
  TK_PROF_ADD(ws, 30);
  TK_PROF_MARK(pq0_);
  int cbp = quantize_team<SC>(t, ws, coefq, qp, size, (coeff_type >> 1) & 1);
  TK_PROF_ACC(ws, 12, pq0_);
  if (cbp) {
    TK_PROF_MARK(pi0_);
    dequantize<SC>(t, ws, coefq, qp, size);
    inv_transform_recon<PIX, SP>(t, ws, pred, pstride, rec, rstride, size, bitdepth);
    TK_PROF_ACC(ws, 31, pi0_);
  } else {
    copy_block<SP, SP>(t, rec, rstride, pred, pstride, size, size);
    t.sync();
  }
#if !defined(THOR_PROF_ME) && !defined(THOR_PROF_MD)   // -DTHOR_PROF_ME / _MD: slots 16..25 hold the motion-search cycles by block size / the work-queue items by kind instead
  TK_PROF_ADD(ws, (size <= 4 ? 16 : size == 8 ? 17 : size == 16 ? 18 : size == 32 ? 19 : 20));
  TK_PROF_CNT(ws, (size <= 4 ? 21 : size == 8 ? 22 : size == 16 ? 23 : size == 32 ? 24 : 25));
#endif
  TK_PROF_ADD(ws, 6);
  return cbp;
}
// Entry point for callers that do not know the spaces at compile time (KAT kernels): wave-uniform run-time selection.
template <typename PIX>
TK_DEV int code_tu(const Team t, XformWs* ws, const PIX* org, int ostride, const PIX* pred, int pstride, PIX* rec,
                   int rstride, int size, int qp, int coeff_type, int fast, int16_t* coefq, int bitdepth) {
  const int lb = tk_uniform(tk_is_lds(pred)), lc = tk_uniform(tk_is_lds(coefq));
  if (lb) return code_tu_sp<PIX, SP_LDS, SP_LDS>(t, ws, org, ostride, pred, pstride, rec, rstride, size, qp, coeff_type, fast, coefq, bitdepth);
  if (lc) return code_tu_sp<PIX, SP_GLOBAL, SP_LDS>(t, ws, org, ostride, pred, pstride, rec, rstride, size, qp, coeff_type, fast, coefq, bitdepth);
  return code_tu_sp<PIX, SP_GLOBAL, SP_GLOBAL>(t, ws, org, ostride, pred, pstride, rec, rstride, size, qp, coeff_type, fast, coefq, bitdepth);
}

}  // namespace tk
