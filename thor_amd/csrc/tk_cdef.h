// tk_cdef.h - Constrained Directional Enhancement Filter: encoder-side strength search + frame
// filtering, work-item parallel (in-loop and ON in every *_high_efficiency
// config, so bit-exact streams need it).
// Specification followed: enc/encode_frame.c:47-486 (search_one_dual / joint_strength_search_dual,
// dist_8x8, cdef_search), common/common_frame.c:766-1003 (cdef_prepare_input, cdef_allskip,
// cdef_frame), common/common_block.c:94-279 (cdef_find_dir, direction tables, constrain,
// cdef_filter_block), common/common_frame.h:61-72 (adjust_strength, cdef_init).
// The reference filters in place through a write-back cache that guarantees every neighbour it
// reads is still unfiltered; here the deblocked frame is copied to `src` and filtered out of place,
// which is the same function.  Frame-outside taps read CDEF_VERY_LARGE exactly as
// cdef_prepare_input produces them.  4:2:0 only.
#pragma once
#include "tk_common.h"

namespace tk {

enum { kCdefVeryLarge = 30000, kCdefMaxStr = 64 };

struct CdefResult {
  int nb_bits;                 // bits per filter block actually signalled
  int strengths[8], uv_strengths[8];   // header values (pri*4 + sec after priconv)
  int sb_count;                // number of non-all-skip filter blocks
  int level_tab[2][8];         // scratch
};

template <typename PIX> struct CdefJob {
  Plane3<PIX> rec;   // output (in-loop filtered)
  Plane3<PIX> src;   // deblocked input copy
  Plane3<PIX> org;
  int width, height, bitdepth;
  const DbCell* cells;
  int cs;
  int nfb_h, nfb_v;
  int speed, damping, cdef_bits;   // speed = cdef-1; cdef_bits = header guess
  int qp;
  int8_t* dir;                     // per 8x8 luma block
  int* var;
  int* fb_compact;                 // per fb: compact index or -1 (all skip)
  unsigned long long* mse;         // [2][nfb][64]
  int* sel;                        // per compact index: selected preset
  int* fb_sel;                     // per fb: selected preset (or 0)
  CdefResult* res;
  unsigned long long* tot;         // [64*64] scratch for the joint search + [nfb] (what the chosen pairs give each non-skip filter block)
};

TK_DEV int cdef_priconv(int speed, int i) {
  if (speed == 0) return i;
  if (speed == 1) { const int t[8] = {0, 1, 2, 3, 5, 7, 10, 13}; return t[i & 7]; }
  const int t[4] = {0, 1, 3, 6};
  return t[i & 3];
}
TK_HD int cdef_total_strengths(int speed) { return speed == 0 ? 64 : (speed == 1 ? 32 : 16); }

TK_DEV int cdef_dx(int d, int k) {
  const int8_t t[8][2] = {{1, 2}, {1, 2}, {1, 2}, {1, 2}, {1, 2}, {0, 1}, {0, 0}, {0, -1}};
  return t[d][k];
}
TK_DEV int cdef_dy(int d, int k) {
  const int8_t t[8][2] = {{-1, -2}, {0, -1}, {0, 0}, {0, 1}, {1, 2}, {1, 2}, {1, 2}, {1, 2}};
  return t[d][k];
}

TK_DEV int cdef_constrain(int diff, int threshold, int damping) {
  if (!threshold) return 0;
  int a = iabs(diff);
  int m = threshold - (a >> (damping - ilog2((unsigned)threshold)));
  if (m < 0) m = 0;
  int r = a < m ? a : m;
  return diff < 0 ? -r : r;
}

template <typename PIX> TK_DEV int cdef_fetch(const PIX* p, int stride, int x, int y, int w, int h) {
  if (x < 0 || y < 0 || x >= w || y >= h) return kCdefVeryLarge;
  return p[y * stride + x];
}

// cdef_filter_block for one sample (common_block.c:224-279)
template <typename PIX>
TK_DEV int cdef_filter_px(const PIX* p, int stride, int x, int y, int w, int h, int pri, int sec, int dir,
                          int pri_damp, int sec_damp, int cs) {
  const int sel = (pri >> cs) & 1;
  const int X = p[y * stride + x];
  int sum = 0, mx = X, mn = X;
  for (int k = 0; k < 2; k++) {
    const int pt = sel ? 3 : (k == 0 ? 4 : 2);
    const int st = k == 0 ? 2 : 1;
    int ox = cdef_dx(dir, k), oy = cdef_dy(dir, k);
    int p0 = cdef_fetch(p, stride, x + ox, y + oy, w, h), p1 = cdef_fetch(p, stride, x - ox, y - oy, w, h);
    sum = (int16_t)(sum + pt * cdef_constrain(p0 - X, pri, pri_damp));
    sum = (int16_t)(sum + pt * cdef_constrain(p1 - X, pri, pri_damp));
    if (p0 != kCdefVeryLarge) mx = tmax(p0, mx);
    if (p1 != kCdefVeryLarge) mx = tmax(p1, mx);
    mn = tmin(p0, mn); mn = tmin(p1, mn);
    int d2 = (dir + 2) & 7, d6 = (dir + 6) & 7;
    int ax = cdef_dx(d2, k), ay = cdef_dy(d2, k), bx = cdef_dx(d6, k), by = cdef_dy(d6, k);
    int s0 = cdef_fetch(p, stride, x + ax, y + ay, w, h), s1 = cdef_fetch(p, stride, x - ax, y - ay, w, h);
    int s2 = cdef_fetch(p, stride, x + bx, y + by, w, h), s3 = cdef_fetch(p, stride, x - bx, y - by, w, h);
    if (s0 != kCdefVeryLarge) mx = tmax(s0, mx);
    if (s1 != kCdefVeryLarge) mx = tmax(s1, mx);
    if (s2 != kCdefVeryLarge) mx = tmax(s2, mx);
    if (s3 != kCdefVeryLarge) mx = tmax(s3, mx);
    mn = tmin(s0, mn); mn = tmin(s1, mn); mn = tmin(s2, mn); mn = tmin(s3, mn);
    sum = (int16_t)(sum + st * cdef_constrain(s0 - X, sec, sec_damp));
    sum = (int16_t)(sum + st * cdef_constrain(s1 - X, sec, sec_damp));
    sum = (int16_t)(sum + st * cdef_constrain(s2 - X, sec, sec_damp));
    sum = (int16_t)(sum + st * cdef_constrain(s3 - X, sec, sec_damp));
  }
  int yv = (int16_t)(X + ((8 + sum - (sum < 0)) >> 4));
  yv = yv < mn ? mn : yv;   // clip(n, low, high) = min(high, max(n, low))
  yv = yv > mx ? mx : yv;
  return yv;
}

TK_DEV int cdef_adjust_strength(int strength, int var) {
  const int i = (var >> 6) ? tmin(ilog2((unsigned)(var >> 6)), 12) : 0;
  return var ? (strength * (4 + i) + 8) >> 4 : 0;
}

// cdef_find_dir (common_block.c:94-162) on an 8x8 block
template <typename PIX> TK_DEV int cdef_find_dir(const PIX* img, int stride, int* var, int cs) {
  int cost[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int partial[8][15];
  for (int a = 0; a < 8; a++)
    for (int b = 0; b < 15; b++) partial[a][b] = 0;
  const int div_table[9] = {0, 840, 420, 280, 210, 168, 140, 120, 105};
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) {
      int x = (img[i * stride + j] >> cs) - 128;
      partial[0][i + j] += x;
      partial[1][i + j / 2] += x;
      partial[2][i] += x;
      partial[3][3 + i - j / 2] += x;
      partial[4][7 + i - j] += x;
      partial[5][3 - i / 2 + j] += x;
      partial[6][j] += x;
      partial[7][i / 2 + j] += x;
    }
  for (int i = 0; i < 8; i++) {
    cost[2] += partial[2][i] * partial[2][i];
    cost[6] += partial[6][i] * partial[6][i];
  }
  cost[2] *= div_table[8];
  cost[6] *= div_table[8];
  for (int i = 0; i < 7; i++) {
    cost[0] += (partial[0][i] * partial[0][i] + partial[0][14 - i] * partial[0][14 - i]) * div_table[i + 1];
    cost[4] += (partial[4][i] * partial[4][i] + partial[4][14 - i] * partial[4][14 - i]) * div_table[i + 1];
  }
  cost[0] += partial[0][7] * partial[0][7] * div_table[8];
  cost[4] += partial[4][7] * partial[4][7] * div_table[8];
  for (int i = 1; i < 8; i += 2) {
    for (int j = 0; j < 5; j++) cost[i] += partial[i][3 + j] * partial[i][3 + j];
    cost[i] *= div_table[8];
    for (int j = 0; j < 3; j++)
      cost[i] += (partial[i][j] * partial[i][j] + partial[i][10 - j] * partial[i][10 - j]) * div_table[2 * j + 2];
  }
  int best_cost = 0, best_dir = 0;
  for (int i = 0; i < 8; i++)
    if (cost[i] > best_cost) { best_cost = cost[i]; best_dir = i; }
  *var = (best_cost - cost[(best_dir + 4) & 7]) >> 10;
  return best_dir;
}

TK_DEV int cdef_fb_allskip(const DbCell* cells, int cs, int xoff, int yoff, int width, int height) {
  for (int m = 0; m < 8; m++)
    for (int n = 0; n < 8; n++) {
      int xpos = xoff + n * 8, ypos = yoff + m * 8;
      if (xpos < width && ypos < height && cells[(ypos / 4) * cs + xpos / 4].mode != M_SKIP) return 0;
    }
  return 1;
}

// ---- pass 0: per filter block flags; zero mse ------------------------------------------------
template <typename PIX> TK_DEV void cdef_pass_flags(const CdefJob<PIX>& J, int gid, int gsize) {
  const int nfb = J.nfb_h * J.nfb_v;
  for (int fb = gid; fb < nfb; fb += gsize) {
    int k = fb / J.nfb_h, l = fb - k * J.nfb_h;
    J.fb_compact[fb] = cdef_fb_allskip(J.cells, J.cs, l * 64, k * 64, J.width, J.height) ? -1 : 0;
    J.fb_sel[fb] = 0;
  }
  for (int i = gid; i < 2 * nfb * kCdefMaxStr; i += gsize) J.mse[i] = 0;
}

// ---- pass 1: direction / variance per 8x8 luma block of non-skip filter blocks -------------
template <typename PIX> TK_DEV void cdef_pass_dir(const CdefJob<PIX>& J, int gid, int gsize) {
  const int bw = J.width / 8, bh = J.height / 8;
  for (int b = gid; b < bw * bh; b += gsize) {
    int by = b / bw, bx = b - by * bw;
    int fb = (by / 8) * J.nfb_h + bx / 8;
    if (J.fb_compact[fb] < 0) continue;
    int v;
    J.dir[b] = (int8_t)cdef_find_dir(J.src.y + by * 8 * J.src.sy + bx * 8, J.src.sy, &v, J.bitdepth - 8);
    J.var[b] = v;
  }
}

// dist_8x8 (encode_frame.c:194-221): perceptually weighted 8x8 distortion, double arithmetic.  The five sums are accumulated sample by sample while
// the block is filtered (round 6: the 64 filtered samples used to sit in a private array - 400 bytes of scratch per lane, a store and a load per sample).
struct CdefDist {
  unsigned long long sum_s = 0, sum_d = 0, sum_s2 = 0, sum_d2 = 0, sum_sd = 0;
};
TK_DEV void cdef_dist_add(CdefDist& a, unsigned s_, unsigned d_) {
  const unsigned long long s = s_, d = d_;
  a.sum_s += s; a.sum_d += d; a.sum_s2 += s * s; a.sum_d2 += d * d; a.sum_sd += s * d;
}
TK_DEV unsigned long long cdef_dist_finish(const CdefDist& a, int cs) {
  unsigned long long svar = a.sum_s2 - ((a.sum_s * a.sum_s + 32) >> 6);
  unsigned long long dvar = a.sum_d2 - ((a.sum_d * a.sum_d + 32) >> 6);
  double num = (double)(a.sum_d2 + a.sum_s2 - 2 * a.sum_sd) * .5;
  num = num * (double)(svar + dvar + (unsigned long long)(400 << (2 * cs)));
  double den = sqrt((double)(20000 << (4 * cs)) + (double)svar * (double)dvar);
  return (unsigned long long)floor(.5 + num / den);
}

// ---- pass 2: mse[plane group][fb][gi] (encode_frame.c:286-378) ------------------------------
// work item = (8x8 luma-unit block, gi); each item handles Y, U and V of its block.
template <typename PIX> TK_DEV void cdef_pass_mse(const CdefJob<PIX>& J, int gid, int gsize) {
  const int bw = J.width / 8, bh = J.height / 8;
  const int total = cdef_total_strengths(J.speed);
  const int nfb = J.nfb_h * J.nfb_v;
  const int cs = J.bitdepth - 8;
  for (int it = gid; it < bw * bh * total; it += gsize) {
    int b = it / total, gi = it - b * total;
    int by = b / bw, bx = b - by * bw;
    int fb = (by / 8) * J.nfb_h + bx / 8;
    if (J.fb_compact[fb] < 0) continue;
    if (J.cells[(by * 2) * J.cs + bx * 2].mode == M_SKIP) continue;
    const int pri = cdef_priconv(J.speed, gi / 4), sec = gi % 4;
    // luma
    {
      const int adj = cdef_adjust_strength(pri, J.var[b]);
      const int pd = adj ? tmax(ilog2((unsigned)adj), J.damping) : J.damping;
      const int dir = pri ? J.dir[b] : 0;
      CdefDist acc;
      const PIX* const org = J.org.y + by * 8 * J.org.sy + bx * 8;
      for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++)
          cdef_dist_add(acc, org[i * J.org.sy + j], (unsigned)cdef_filter_px(J.src.y, J.src.sy, bx * 8 + j, by * 8 + i, J.width, J.height, adj << cs, sec << cs,
                                                                            dir, pd + cs, J.damping + cs, cs));
      unsigned long long d = cdef_dist_finish(acc, cs);
      team_add64(&J.mse[(0 * nfb + fb) * kCdefMaxStr + gi], d);
    }
    // chroma (encode_frame.c:295-372 with bs = 8 for every plane): the search walks CHROMA 8x8 blocks
    // (m, n) of the filter block but takes the skip flag and the direction from the LUMA 8x8 block
    // with the same (m, n) - i.e. from this work item's block - and covers only the first
    // (h+7)>>4 x (w+7)>>4 of them.  Reproduced as is (cdef_frame later filters co-located 4x4s).
    {
      const int fbx = bx / 8, fby = by / 8, n = bx & 7, m = by & 7;
      int h = tmin(J.height, (fby + 1) << 6) & 63, w = tmin(J.width, (fbx + 1) << 6) & 63;
      h += (!h) << 6;
      w += (!w) << 6;
      if (m < ((h + 7) >> 4) && n < ((w + 7) >> 4)) {
        const int cx = fbx * 32 + n * 8, cy = fby * 32 + m * 8;
        const int sizex = tmin(J.width / 2 - cx, 8), sizey = tmin(J.height / 2 - cy, 8);
        const int pd = pri ? tmax(ilog2((unsigned)pri), J.damping - 1) : J.damping - 1;
        const int dir = pri ? J.dir[b] : 0;
        unsigned long long sse = 0;
        for (int pl = 0; pl < 2; pl++) {
          const PIX* s = pl ? J.src.v : J.src.u;
          const PIX* o = pl ? J.org.v : J.org.u;
          for (int i = 0; i < sizey; i++)
            for (int j = 0; j < sizex; j++) {
              int v = cdef_filter_px(s, J.src.sc, cx + j, cy + i, J.width / 2, J.height / 2, pri << cs, sec << cs, dir,
                                     pd + cs, J.damping - 1 + cs, cs);
              int e = v - (int)o[(cy + i) * J.org.sc + cx + j];
              sse += (unsigned long long)(long long)(e * e);
            }
        }
        team_add64(&J.mse[(1 * nfb + fb) * kCdefMaxStr + gi], sse);
      }
    }
  }
}

// ---- pass 2, wavefront form (round 6) ---------------------------------------------------------
// cdef_pass_mse above is the plain statement: one lane per (block, strength), every lane filtering all the block's samples by itself - 12 bounds-checked
// loads and 12 constrain() per sample and strength.  Per sample, though, the taps, their differences to the centre and the clamp range depend on the
// direction only, the primary sum on (direction, primary strength) only and the secondary sum on (direction, secondary strength) only: the 64 strengths of a
// block need 15 primary sums and 2 x 3 secondary sums per sample (strengths with primary 0 filter along direction 0, common_block.c / encode_frame.c:318),
// not 64 x 12 constrain().  So here ONE WAVEFRONT takes a block:
//   load:   the block with its 2-sample halo into LDS once (12 x 12 int16, kCdefVeryLarge outside the frame), the original block beside it;
//   filter: lane = SAMPLE.  Two tap sets (direction 0, the block's direction), the sums above, then the 64 filtered values of the sample, written to an LDS
//           matrix [strength][sample] (luma: the value; chroma: |value - original|);
//   sums:   lane = STRENGTH.  Its row of 64 samples as packed dot products (v_dot4_u32_u8 / v_dot2_u32_u16): sum d, sum d^2, sum s*d -> dist_8x8 in double
//           arithmetic exactly as before; chroma: sum e^2.  One atomic add per strength.
// Integer results identical to the plain form by construction (no (int16_t) wrap can occur: |sum| <= 12 * 240 + 12 * 48, samples < 4096);
// tests/hostsim/unit_cdef.cpp compares the two on random frames (edges, skipped blocks, partial filter blocks, 8/10/12 bits, the three speeds), and the
// host simulation runs THIS form (lanes as a loop), so every golden stream with cdef on checks it.
template <typename PIX> struct CdefWaveWs {
  enum { kRow = 64 * (int)sizeof(PIX) / 4 + 1 };   // dwords per strength row: 64 samples + one pad dword (lane = strength reads stride kRow: conflict-free)
  int16_t tile[12 * 12];
  uint32_t org[64 * sizeof(PIX) / 4];
  uint32_t rows[64 * kRow];
};
TK_DEV unsigned cdef_udot4(unsigned a, unsigned b, unsigned c) {
#if TK_HOST
  for (int k = 0; k < 4; k++) c += ((a >> (8 * k)) & 0xffu) * ((b >> (8 * k)) & 0xffu);
  return c;
#else
  return __builtin_amdgcn_udot4(a, b, c, false);
#endif
}
TK_DEV unsigned cdef_udot2(unsigned a, unsigned b, unsigned c) {
#if TK_HOST
  return c + (a & 0xffffu) * (b & 0xffffu) + (a >> 16) * (b >> 16);
#else
  typedef unsigned short __attribute__((ext_vector_type(2))) u16x2;
  u16x2 x, y;
  __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4);
  return __builtin_amdgcn_udot2(x, y, c, false);
#endif
}
// constrain() with the threshold's shift taken out (wave-uniform): thr > 0, sh = damping - ilog2(thr)
TK_DEV int cdef_constrain_sh(int diff, int thr, int sh) {
  const int a = iabs(diff);
  const int m = tmax(thr - (a >> sh), 0);
  const int r = tmin(a, m);
  return diff < 0 ? -r : r;
}
// load (one lane's share): tile entries lane, lane + 64, lane + 128 and original sample `lane`; (x0, y0) = the block's position in a plane of w x h samples,
// sx x sy = the part of the 8x8 block inside the plane (chroma blocks at the right / bottom edge)
template <typename PIX>
TK_DEV void cdef_wave_load(CdefWaveWs<PIX>* ws, int lane, const PIX* src, int sstride, const PIX* org, int ostride, int x0, int y0, int w, int h, int sx, int sy) {
  for (int e = lane; e < 144; e += 64) {
    const int ty = e / 12, tx = e - ty * 12;
    ldsc(ws->tile)[e] = (int16_t)cdef_fetch(src, sstride, x0 + tx - 2, y0 + ty - 2, w, h);
  }
  const int i = lane >> 3, j = lane & 7;
  ((TK_LDS PIX*)ldsc(ws->org))[lane] = (i < sy && j < sx) ? org[(y0 + i) * ostride + x0 + j] : (PIX)0;
}
// tap k (0: near, 1: far) along direction d as an offset in the 12-wide tile: cdef_dy(d, k) * 12 + cdef_dx(d, k), from packed bytes biased by 32 (d is
// wave-uniform: scalar shifts instead of a table in memory)
TK_DEV int cdef_off12(int d, int k) {
  const unsigned long long t = k ? 0x3738393a2e22160aull : 0x2c2c2c2d21212115ull;
  return (int)((t >> (8 * d)) & 0xffu) - 32;
}
// the twelve taps of a sample along direction `dir`: primary differences pd[4] (+near, -near, +far, -far), the clamp range, the secondary sums of sec = 1..3
TK_DEV void cdef_wave_taps(const TK_LDS int16_t* c, int X, int dir, int cs, int sec_damp, int* pd, int& mn, int& mx, int* S) {
  const int d2 = (dir + 2) & 7, d6 = (dir + 6) & 7;
  int lo = X, hi = X;
  int sd[2][4];
  for (int k = 0; k < 2; k++) {
    const int po = cdef_off12(dir, k), ao = cdef_off12(d2, k), bo = cdef_off12(d6, k);
    const int t[6] = {c[po], c[-po], c[ao], c[-ao], c[bo], c[-bo]};
    for (int q = 0; q < 6; q++) {
      lo = tmin(lo, t[q]);
      hi = tmax(hi, t[q] == kCdefVeryLarge ? X : t[q]);
    }
    pd[2 * k] = t[0] - X; pd[2 * k + 1] = t[1] - X;
    for (int q = 0; q < 4; q++) sd[k][q] = t[2 + q] - X;
  }
  mn = lo; mx = hi;
  for (int sec = 1; sec < 4; sec++) {
    const int thr = sec << cs, sh = sec_damp - ilog2((unsigned)thr);
    int s = 0;
    for (int q = 0; q < 4; q++) s += 2 * cdef_constrain_sh(sd[0][q], thr, sh) + cdef_constrain_sh(sd[1][q], thr, sh);
    S[sec - 1] = s;
  }
}
// filter (one lane = one sample).  CHROMA: thresholds from the primary strength itself (no variance adjustment), damping - 1, and |filtered - original|
// into the matrix (0 outside sx x sy).
template <typename PIX, bool CHROMA>
TK_DEV void cdef_wave_filter(CdefWaveWs<PIX>* ws, int lane, int speed, int var, int dirb, int damping, int cs, int sx, int sy) {
  const int npri = cdef_total_strengths(speed) / 4;
  const int i = lane >> 3, j = lane & 7;
  const TK_LDS int16_t* const c = ldsc(ws->tile) + (i + 2) * 12 + (j + 2);
  const int X = c[0];
  const int damp = CHROMA ? damping - 1 : damping;
  // [0]: direction 0 (strengths with primary 0), [1]: the block's direction
  int pd0[4], pd[4], mn0, mx0, mn1, mx1, S0[3], S1[3];
  cdef_wave_taps(c, X, 0, cs, damp + cs, pd0, mn0, mx0, S0);
  if (dirb) cdef_wave_taps(c, X, dirb, cs, damp + cs, pd, mn1, mx1, S1);   // wave-uniform branch
  else {
    mn1 = mn0; mx1 = mx0;
    for (int q = 0; q < 3; q++) S1[q] = S0[q];
    for (int q = 0; q < 4; q++) pd[q] = pd0[q];
  }
  const int O = CHROMA ? (int)((const TK_LDS PIX*)ldsc(ws->org))[lane] : 0;
  const bool inside = !CHROMA || (i < sy && j < sx);
  TK_LDS PIX* const out = (TK_LDS PIX*)ldsc(ws->rows) + lane;
  const int rowpix = CdefWaveWs<PIX>::kRow * 4 / (int)sizeof(PIX);
  for (int k = 0; k < npri; k++) {
    const int pri = cdef_priconv(speed, k);
    const int str = CHROMA ? pri : cdef_adjust_strength(pri, var);
    const int thr = str << cs;
    const int v = pri ? 1 : 0;
    int P = 0;
    if (thr) {   // wave-uniform
      const int sh = tmax(ilog2((unsigned)str), damp) + cs - ilog2((unsigned)thr);
      const int w0 = (str & 1) ? 3 : 4, w1 = (str & 1) ? 3 : 2;
      P = w0 * (cdef_constrain_sh(pd[0], thr, sh) + cdef_constrain_sh(pd[1], thr, sh)) + w1 * (cdef_constrain_sh(pd[2], thr, sh) + cdef_constrain_sh(pd[3], thr, sh));
    }
    for (int sec = 0; sec < 4; sec++) {
      const int sum = P + (sec ? (v ? S1[sec ? sec - 1 : 0] : S0[sec ? sec - 1 : 0]) : 0);
      int y = X + ((8 + sum - (sum < 0)) >> 4);
      const int lo = v ? mn1 : mn0, hi = v ? mx1 : mx0;
      y = y < lo ? lo : y;
      y = y > hi ? hi : y;
      out[(k * 4 + sec) * rowpix] = (PIX)(CHROMA ? (inside ? iabs(y - O) : 0) : y);
    }
  }
}
// sums (one lane = one strength gi): luma -> dist_8x8 of the filtered block against the original; chroma -> sum of squared errors
template <typename PIX, bool CHROMA> TK_DEV unsigned long long cdef_wave_sums(const CdefWaveWs<PIX>* ws, int gi, int cs) {
  const TK_LDS uint32_t* const row = ldsc(ws->rows) + gi * CdefWaveWs<PIX>::kRow;
  const TK_LDS uint32_t* const org = ldsc(ws->org);
  unsigned sd = 0, sd2 = 0, ssd = 0, ss = 0, ss2 = 0;
  const unsigned ones = sizeof(PIX) == 1 ? 0x01010101u : 0x00010001u;
  for (int q = 0; q < 64 * (int)sizeof(PIX) / 4; q++) {
    const unsigned d = row[q];
    if constexpr (sizeof(PIX) == 1) {
      sd2 = cdef_udot4(d, d, sd2);
      if (!CHROMA) { const unsigned s = org[q]; sd = cdef_udot4(d, ones, sd); ssd = cdef_udot4(d, s, ssd); ss = cdef_udot4(s, ones, ss); ss2 = cdef_udot4(s, s, ss2); }
    } else {
      sd2 = cdef_udot2(d, d, sd2);
      if (!CHROMA) { const unsigned s = org[q]; sd = cdef_udot2(d, ones, sd); ssd = cdef_udot2(d, s, ssd); ss = cdef_udot2(s, ones, ss); ss2 = cdef_udot2(s, s, ss2); }
    }
  }
  if (CHROMA) return sd2;
  CdefDist a;
  a.sum_s = ss; a.sum_d = sd; a.sum_s2 = ss2; a.sum_d2 = sd2; a.sum_sd = ssd;
  return cdef_dist_finish(a, cs);
}
// The lanes of the wavefront: the hardware's on the device (t.rank), a loop on the host.
#if TK_HOST
#define TK_CDEF_LANES(l) for (int l = 0; l < 64; l++)
#else
#define TK_CDEF_LANES(l) for (int l = t.rank, once_ = 1; once_; once_ = 0)
#endif
// One 8x8 luma-unit block b (all of the team's lanes call this with the same b): the same skips, strengths and sums as one b of cdef_pass_mse.
template <typename PIX> TK_DEV void cdef_mse_block_wave(const Team t, const CdefJob<PIX>& J, int b, CdefWaveWs<PIX>* ws) {
  const int bw = J.width / 8;
  const int total = cdef_total_strengths(J.speed);
  const int nfb = J.nfb_h * J.nfb_v;
  const int cs = J.bitdepth - 8;
  const int by = b / bw, bx = b - by * bw;
  const int fb = (by / 8) * J.nfb_h + bx / 8;
  if (J.fb_compact[fb] < 0) return;
  if (J.cells[(by * 2) * J.cs + bx * 2].mode == M_SKIP) return;
  const int var = J.var[b], dirb = J.dir[b];
  // luma
  TK_CDEF_LANES(l) cdef_wave_load(ws, l, J.src.y, J.src.sy, J.org.y, J.org.sy, bx * 8, by * 8, J.width, J.height, 8, 8);
  t.sync();
  TK_CDEF_LANES(l) cdef_wave_filter<PIX, false>(ws, l, J.speed, var, dirb, J.damping, cs, 8, 8);
  t.sync();
  TK_CDEF_LANES(l) if (l < total) team_add64(&J.mse[(0 * nfb + fb) * kCdefMaxStr + l], cdef_wave_sums<PIX, false>(ws, l, cs));
  // chroma: the (m, n)-th 8x8 CHROMA block of the filter block with this luma block's skip flag and direction, the first (h+7)>>4 x (w+7)>>4 only (see above)
  const int fbx = bx / 8, fby = by / 8, n = bx & 7, m = by & 7;
  int h = tmin(J.height, (fby + 1) << 6) & 63, w = tmin(J.width, (fbx + 1) << 6) & 63;
  h += (!h) << 6;
  w += (!w) << 6;
  if (m < ((h + 7) >> 4) && n < ((w + 7) >> 4)) {
    const int cx = fbx * 32 + n * 8, cy = fby * 32 + m * 8;
    const int sizex = tmin(J.width / 2 - cx, 8), sizey = tmin(J.height / 2 - cy, 8);
    unsigned long long sse[TK_HOST ? 64 : 1];
    TK_CDEF_LANES(l) sse[TK_HOST ? l : 0] = 0;
    for (int pl = 0; pl < 2; pl++) {
      t.sync();   // the previous sums have read the matrix
      TK_CDEF_LANES(l) cdef_wave_load(ws, l, pl ? J.src.v : J.src.u, J.src.sc, pl ? J.org.v : J.org.u, J.org.sc, cx, cy, J.width / 2, J.height / 2, sizex, sizey);
      t.sync();
      TK_CDEF_LANES(l) cdef_wave_filter<PIX, true>(ws, l, J.speed, var, dirb, J.damping, cs, sizex, sizey);
      t.sync();
      TK_CDEF_LANES(l) if (l < total) sse[TK_HOST ? l : 0] += cdef_wave_sums<PIX, true>(ws, l, cs);
    }
    TK_CDEF_LANES(l) if (l < total) team_add64(&J.mse[(1 * nfb + fb) * kCdefMaxStr + l], sse[TK_HOST ? l : 0]);
  }
  t.sync();
}

// ---- pass 3: joint luma+chroma strength selection (single team) ------------------------------
// search_one_dual / joint_strength_search_dual (encode_frame.c:86-192) + the sort / dedupe /
// per-block assignment tail of cdef_search (:380-470).
template <typename PIX, class TeamT> TK_DEV void cdef_pass_select(const TeamT t, const CdefJob<PIX>& J) {
  const int nfb = J.nfb_h * J.nfb_v;
  const int total = cdef_total_strengths(J.speed);
  CdefResult* R = J.res;
  // compact list of non-skip filter blocks, raster order (J.sel holds the list during the search; its final contents are written at the end)
  if (t.rank == 0) {
    int n = 0;
    for (int fb = 0; fb < nfb; fb++)
      if (J.fb_compact[fb] >= 0) { J.sel[n] = fb; J.fb_compact[fb] = n++; }
    R->sb_count = n;
  }
  t.block_sync();
  const int sbc = R->sb_count;
  const unsigned long long* mse0 = J.mse;
  const unsigned long long* mse1 = J.mse + (size_t)nfb * kCdefMaxStr;
  unsigned long long* const bestv = J.tot + kCdefMaxStr * kCdefMaxStr;   // [sb_count]
  if (J.cdef_bits == 0) {
    // fixed strengths guessed from the frame QP (encode_frame.c:260-281); no per-block signalling
    if (t.rank == 0) {
      const int pri = tmax(0, (J.qp - 24) / 3), sec = J.qp < 32 && J.qp > 16;
      R->strengths[0] = R->uv_strengths[0] = (pri << 2) + sec;
      R->nb_bits = 0;
    }
    t.block_sync();
    return;
  }
  const int nb_strengths = 1 << J.cdef_bits;
  int* lev0 = R->level_tab[0];
  int* lev1 = R->level_tab[1];
  // greedy + refinement: sequence of search_one_dual calls
  const int ncalls = nb_strengths + 4 * nb_strengths;
  for (int call = 0; call < ncalls; call++) {
    int nsel;
    if (call < nb_strengths) nsel = call;
    else {
      t.block_sync();
      if (t.rank == 0)
        for (int j = 0; j < nb_strengths - 1; j++) { lev0[j] = lev0[j + 1]; lev1[j] = lev1[j + 1]; }
      nsel = nb_strengths - 1;
    }
    t.block_sync();
    // what the already chosen pairs give each filter block does not depend on the candidate pair (j, k): once per call, not once per candidate
    for (int q = t.rank; q < sbc; q += t.size) {
      const int fb = J.sel[q];
      const unsigned long long* m0 = mse0 + (size_t)fb * kCdefMaxStr;
      const unsigned long long* m1 = mse1 + (size_t)fb * kCdefMaxStr;
      unsigned long long best = 1ull << 63;
      for (int g = 0; g < nsel; g++) {
        unsigned long long c = m0[lev0[g]] + m1[lev1[g]];
        if (c < best) best = c;
      }
      bestv[q] = best;
    }
    t.block_sync();
    for (int jk = t.rank; jk < total * total; jk += t.size) {
      const int j = jk / total, k = jk - j * total;
      unsigned long long acc = 0;
      for (int q = 0; q < sbc; q++) {
        const int fb = J.sel[q];
        const unsigned long long c = mse0[(size_t)fb * kCdefMaxStr + j] + mse1[(size_t)fb * kCdefMaxStr + k];
        const unsigned long long best = bestv[q];
        acc += c < best ? c : best;
      }
      J.tot[jk] = acc;
    }
    t.block_sync();
    if (t.rank == 0) {
      unsigned long long bt = 1ull << 63;
      int b0 = 0, b1 = 0;
      for (int j = 0; j < total; j++)
        for (int k = 0; k < total; k++)
          if (J.tot[j * total + k] < bt) { bt = J.tot[j * total + k]; b0 = j; b1 = k; }
      lev0[nsel] = b0;
      lev1[nsel] = b1;
    }
    t.block_sync();
  }
  if (t.rank == 0) {
    int nb_bits = J.cdef_bits;
    int strengths[8], uvs[8];
    for (int j = 0; j < (1 << nb_bits); j++) { strengths[j] = lev0[j]; uvs[j] = lev1[j]; }
    // sort + remove duplicates (qsort on unique keys => any correct ascending sort)
    unsigned list[8];
    int gi_trans[8];
    const int n = 1 << nb_bits;
    for (int i = 0; i < n; i++) list[i] = ((unsigned)strengths[i] << 16) + ((unsigned)uvs[i] << 8) + (unsigned)i;
    for (int i = 1; i < n; i++) {
      unsigned v = list[i];
      int q = i - 1;
      while (q >= 0 && list[q] > v) { list[q + 1] = list[q]; q--; }
      list[q + 1] = v;
    }
    int j = 0;
    for (int i = 0; i < n; i++) {
      gi_trans[list[i] & 255] = j;
      if (!i || (list[i] & ~255u) != (list[i - 1] & ~255u)) {
        strengths[j] = (int)(list[i] >> 16);
        uvs[j++] = (int)((list[i] >> 8) & 255);
      }
    }
    nb_bits = ilog2((unsigned)j);
    const int nbs = 1 << nb_bits;
    for (int fb = 0; fb < nfb; fb++) {
      if (J.fb_compact[fb] < 0) continue;
      const unsigned long long* m0 = mse0 + (size_t)fb * kCdefMaxStr;
      const unsigned long long* m1 = mse1 + (size_t)fb * kCdefMaxStr;
      unsigned long long best = 1ull << 63;
      int best_gi = 0;
      for (int gi = 0; gi < nbs; gi++) {
        unsigned long long c = m0[strengths[gi_trans[gi]]] + m1[uvs[gi_trans[gi]]];
        if (c < best) { best_gi = tmin(nbs - 1, gi_trans[gi]); best = c; }
      }
      J.sel[J.fb_compact[fb]] = best_gi;
      J.fb_sel[fb] = best_gi;
    }
    for (int q = 0; q < nbs; q++) {
      R->strengths[q] = cdef_priconv(J.speed, strengths[q] / 4) * 4 + (strengths[q] % 4);
      R->uv_strengths[q] = cdef_priconv(J.speed, uvs[q] / 4) * 4 + (uvs[q] % 4);
    }
    R->nb_bits = nb_bits;
  }
  t.block_sync();
}

// ---- pass 4: apply (cdef_frame, common_frame.c:826-1003); item = 8x8 luma-unit block --------
template <typename PIX> TK_DEV void cdef_pass_apply(const CdefJob<PIX>& J, int gid, int gsize) {
  const int bw = J.width / 8, bh = J.height / 8;
  const int cs = J.bitdepth - 8;
  const CdefResult* R = J.res;
  for (int b = gid; b < bw * bh; b += gsize) {
    int by = b / bw, bx = b - by * bw;
    int fb = (by / 8) * J.nfb_h + bx / 8;
    if (J.fb_compact[fb] < 0) continue;
    if (J.cells[(by * 2) * J.cs + bx * 2].mode == M_SKIP) continue;
    const int preset = J.fb_sel[fb];
    for (int pg = 0; pg < 2; pg++) {
      const int str = pg ? R->uv_strengths[preset] : R->strengths[preset];
      const int pri = str >> 2;
      int sec = str & 3;
      sec += (sec == 3);
      const int adj = pg ? pri : cdef_adjust_strength(pri, J.var[b]);
      const int pd = adj ? tmax(ilog2((unsigned)adj), J.damping - pg) : J.damping - pg;
      const int sd = J.damping - pg;
      const int dir = pri ? J.dir[b] : 0;
      if (pg == 0) {
        for (int i = 0; i < 8; i++)
          for (int j = 0; j < 8; j++)
            J.rec.y[(by * 8 + i) * J.rec.sy + bx * 8 + j] = (PIX)cdef_filter_px(
                J.src.y, J.src.sy, bx * 8 + j, by * 8 + i, J.width, J.height, adj << cs, sec << cs, dir, pd + cs, sd + cs, cs);
      } else {
        for (int pl = 0; pl < 2; pl++) {
          const PIX* s = pl ? J.src.v : J.src.u;
          PIX* d = pl ? J.rec.v : J.rec.u;
          for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++)
              d[(by * 4 + i) * J.rec.sc + bx * 4 + j] = (PIX)cdef_filter_px(
                  s, J.src.sc, bx * 4 + j, by * 4 + i, J.width / 2, J.height / 2, adj << cs, sec << cs, dir, pd + cs, sd + cs, cs);
        }
      }
    }
  }
}

}  // namespace tk
