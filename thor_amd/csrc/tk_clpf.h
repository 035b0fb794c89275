// tk_clpf.h - constrained low-pass filter (CLPF) of the low/medium-complexity operating points
// (SURVEY.md §8f row 4).  Reference: enc/encode_frame.c:492-635 (clpf_decision, clpf_rdo, clpf_test_frame),
// :785-817 (frame-level signalling), enc/encode_block.c:2556-2621 (detect_clpf / detect_multi_clpf),
// common/common_frame.c:1005-1163 (clpf_frame), common/common_block.c:315-345 (clpf_sample, clpf_block).
//
// Split of the work:
//   device pass 1 (clpf_pass_stats): per 8x8 block of every plane the squared errors against the original
//       for "unfiltered" and strengths 1, 2, 4 (exactly the per-block terms detect_multi_clpf adds up);
//   host (clpf_plan): the frame-level choice - hierarchical filter-block RDO 128/64/32 for luma, strength per
//       plane, the per-filter-block on/off bits - which is a few thousand integer additions;
//   device pass 2 (clpf_pass_apply): the filter itself, reading the unfiltered copy (the reference's
//       write-back cache makes its in-place filter semantically out-of-place, common_frame.c:1030-1140).
#pragma once
#include <cstring>
#include <utility>
#include <vector>
#include "tk_common.h"
#include "tk_cdef.h"

namespace tk {

enum { kClpfSkipMark = 0xffffffffu };

template <typename PIX> struct ClpfJob {
  Plane3<PIX> rec;   // filtered output
  Plane3<PIX> src;   // unfiltered copy (pass 2 input)
  Plane3<PIX> org;
  int width, height, bitdepth, qp;
  const DbCell* cells;
  int cs;
  uint32_t* stats;       // [nblk_y + 2 * nblk_c][4]; [1] == kClpfSkipMark for skip-coded blocks
  // pass 2 parameters (filled by the host plan)
  int strength[3];       // 0 = plane not filtered; else 1, 2 or 4 (before the bitdepth shift)
  int fb_log2;           // luma filter-block size
  const uint8_t* fb_on;  // per luma filter block: 1 = filter (flag or "no flags" mode), 0 = leave
};

TK_DEV int clpf_sample(int X, int A, int B, int C, int D, int E, int F, int G, int H, int s, int dmp) {
  int delta = 1 * cdef_constrain(A - X, s, dmp) + 3 * cdef_constrain(B - X, s, dmp) + 1 * cdef_constrain(C - X, s, dmp) +
              3 * cdef_constrain(D - X, s, dmp) + 3 * cdef_constrain(E - X, s, dmp) + 1 * cdef_constrain(F - X, s, dmp) +
              3 * cdef_constrain(G - X, s, dmp) + 1 * cdef_constrain(H - X, s, dmp);
  return (8 + delta - (delta < 0)) >> 4;
}

template <typename PIX> struct ClpfPlane {
  const PIX* rec; const PIX* org; PIX* dst;
  int w, h, rs, os, ds;
};
template <typename PIX> TK_DEV ClpfPlane<PIX> clpf_plane(const ClpfJob<PIX>& J, int plane, bool from_src) {
  ClpfPlane<PIX> P;
  const Plane3<PIX>& r = from_src ? J.src : J.rec;
  if (plane == 0) { P.rec = r.y; P.org = J.org.y; P.dst = J.rec.y; P.w = J.width; P.h = J.height; P.rs = r.sy; P.os = J.org.sy; P.ds = J.rec.sy; }
  else {
    P.rec = plane == 1 ? r.u : r.v; P.org = plane == 1 ? J.org.u : J.org.v; P.dst = plane == 1 ? J.rec.u : J.rec.v;
    P.w = J.width / 2; P.h = J.height / 2; P.rs = r.sc; P.os = J.org.sc; P.ds = J.rec.sc;
  }
  return P;
}

// pass 1: one work item per 8x8 block (of any plane).  Block order: luma raster, then U, then V.
template <typename PIX> TK_DEV void clpf_pass_stats(const ClpfJob<PIX>& J, int gid, int gsize) {
  const int nby = (J.width / 8) * (J.height / 8), nbc = (J.width / 16) * (J.height / 16);
  const int shift = J.bitdepth - 8;
  for (int b = gid; b < nby + 2 * nbc; b += gsize) {
    const int plane = b < nby ? 0 : (b < nby + nbc ? 1 : 2);
    const int bi = plane == 0 ? b : (plane == 1 ? b - nby : b - nby - nbc);
    const ClpfPlane<PIX> P = clpf_plane(J, plane, false);
    const int bw = P.w / 8;
    const int x0 = (bi % bw) * 8, y0 = (bi / bw) * 8;
    const int sub = plane != 0;
    // skip test on the cell holding the block's top-left sample (encode_frame.c:586-588)
    const DbCell& cell = J.cells[((y0 << sub) / 4) * J.cs + ((x0 << sub) / 4)];
    uint32_t* out = J.stats + 4 * (size_t)b;
    if (cell.mode == M_SKIP) { out[0] = 0; out[1] = kClpfSkipMark; out[2] = 0; out[3] = 0; continue; }
    const int dmp = J.bitdepth - 4 - (plane != 0) + (J.qp >> 4);
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int y = y0; y < y0 + 8; y++)
      for (int x = x0; x < x0 + 8; x++) {
        const int O = P.org[y * P.os + x];
        const int X = P.rec[y * P.rs + x];
        const int A = P.rec[tmax(0, y - 2) * P.rs + x], B = P.rec[tmax(0, y - 1) * P.rs + x];
        const int C = P.rec[y * P.rs + tmax(0, x - 2)], D = P.rec[y * P.rs + tmax(0, x - 1)];
        const int E = P.rec[y * P.rs + tmin(P.w - 1, x + 1)], F = P.rec[y * P.rs + tmin(P.w - 1, x + 2)];
        const int G = P.rec[tmin(P.h - 1, y + 1) * P.rs + x], H = P.rec[tmin(P.h - 1, y + 2) * P.rs + x];
        const int F1 = X + clpf_sample(X, A, B, C, D, E, F, G, H, 1 << shift, dmp);
        const int F2 = X + clpf_sample(X, A, B, C, D, E, F, G, H, 2 << shift, dmp);
        const int F3 = X + clpf_sample(X, A, B, C, D, E, F, G, H, 4 << shift, dmp);
        s0 += (uint32_t)((O - X) * (O - X)); s1 += (uint32_t)((O - F1) * (O - F1));
        s2 += (uint32_t)((O - F2) * (O - F2)); s3 += (uint32_t)((O - F3) * (O - F3));
      }
    out[0] = s0 >> (shift * 2); out[1] = s1 >> (shift * 2); out[2] = s2 >> (shift * 2); out[3] = s3 >> (shift * 2);
  }
}

// pass 2: one work item per filter unit (8x8 luma, 4x4 chroma).  Unit order: luma raster, U, V.
template <typename PIX> TK_DEV void clpf_pass_apply(const ClpfJob<PIX>& J, int gid, int gsize) {
  const int nuy = (J.width / 8) * (J.height / 8), nuc = (J.width / 8) * (J.height / 8);  // chroma: (w/2/4)*(h/2/4)
  for (int u = gid; u < nuy + 2 * nuc; u += gsize) {
    const int plane = u < nuy ? 0 : (u < nuy + nuc ? 1 : 2);
    if (!J.strength[plane]) continue;
    const int ui = plane == 0 ? u : (plane == 1 ? u - nuy : u - nuy - nuc);
    const ClpfPlane<PIX> P = clpf_plane(J, plane, true);
    const int bs = plane ? 4 : 8;
    const int uw = P.w / bs;
    const int x0 = (ui % uw) * bs, y0 = (ui / uw) * bs;
    const int sub = plane != 0;
    // sic: clpf_frame indexes deblock_data with the PLANE width as row pitch (common_frame.c:1048,1075), so for
    // chroma the skip test reads the cell at linear index (2y/4)*(w/2/4) + 2x/4 of the luma-pitched array.
    const int lin = ((y0 << sub) / 4) * (P.w / 4) + ((x0 << sub) / 4);
    if (J.cells[lin].mode == M_SKIP) continue;
    if (plane == 0) {
      const int nfb_h = (J.width + (1 << J.fb_log2) - 1) >> J.fb_log2;
      if (!J.fb_on[(y0 >> J.fb_log2) * nfb_h + (x0 >> J.fb_log2)]) continue;
    } else {
      // chroma filter blocks are 16x16 chroma samples, filtered unless every unit in them is skip (tested with the
      // same mis-pitched index); a non-skip unit implies its block is not all-skip, so no extra test is needed.
    }
    const int s = J.strength[plane] << (J.bitdepth - 8);
    const int dmp = J.bitdepth - 4 - (plane != 0) + (J.qp >> 4);
    for (int y = y0; y < y0 + bs; y++)
      for (int x = x0; x < x0 + bs; x++) {
        const int X = P.rec[y * P.rs + x];
        const int A = P.rec[tmax(0, y - 2) * P.rs + x], B = P.rec[tmax(0, y - 1) * P.rs + x];
        const int C = P.rec[y * P.rs + tmax(0, x - 2)], D = P.rec[y * P.rs + tmax(0, x - 1)];
        const int E = P.rec[y * P.rs + tmin(P.w - 1, x + 1)], F = P.rec[y * P.rs + tmin(P.w - 1, x + 2)];
        const int G = P.rec[tmin(P.h - 1, y + 1) * P.rs + x], H = P.rec[tmin(P.h - 1, y + 2) * P.rs + x];
        P.dst[y * P.ds + x] = (PIX)(X + clpf_sample(X, A, B, C, D, E, F, G, H, s, dmp));
      }
  }
}

// ---------------------------------------------------------------------------------------------
// host: frame-level plan from the per-block statistics
// ---------------------------------------------------------------------------------------------
struct ClpfPlan {
  int strength[3] = {0, 0, 0};
  int fb_log2 = 7, enable_fb = 0;
  std::vector<uint8_t> fb_on;                         // luma filter blocks, raster
  std::vector<std::pair<int, uint32_t>> bits;         // (nbits, value) in stream order
};

struct ClpfHost {
  const uint32_t* st;  // stats of one plane: [blocks][4]
  int bw;              // blocks per row
  // clpf_rdo (encode_frame.c:523-595); y, x in samples, w, h in 8x8 blocks
  int rdo(int y, int x, int fb_log2, int w, int h, int64_t res[4][4], bool luma) const {
    int filtered = 0;
    if (luma && fb_log2 > 7 - 3) {
      fb_log2--;
      const int half = 1 << (fb_log2 - 3);
      const int w1 = half < w ? half : w, h1 = half < h ? half : h;
      const int w2 = (w - half) < (w >> 1) ? (w - half) : (w >> 1), h2 = (h - half) < (h >> 1) ? (h - half) : (h >> 1);
      const int i = 7 - fb_log2;
      const int64_t sum1 = res[i][1], sum2 = res[i][2], sum3 = res[i][3], oldfiltered = res[i][0];
      res[i][0] = 0;
      filtered = rdo(y, x, fb_log2, w1, h1, res, luma);
      if (half < w) filtered |= rdo(y, x + (1 << fb_log2), fb_log2, w2, h1, res, luma);
      if (half < h) {
        filtered |= rdo(y + (1 << fb_log2), x, fb_log2, w1, h2, res, luma);
        filtered |= rdo(y + (1 << fb_log2), x + (1 << fb_log2), fb_log2, w2, h2, res, luma);
      }
      res[i][1] = sum1 + res[i][0] < res[i][1] ? sum1 + res[i][0] : res[i][1];
      res[i][2] = sum2 + res[i][0] < res[i][2] ? sum2 + res[i][0] : res[i][2];
      res[i][3] = sum3 + res[i][0] < res[i][3] ? sum3 + res[i][0] : res[i][3];
      res[i][0] = oldfiltered + filtered;
      return filtered;
    }
    int sum[4] = {0, 0, 0, 0};
    for (int m = 0; m < h; m++)
      for (int n = 0; n < w; n++) {
        const uint32_t* b = st + 4 * (size_t)((y / 8 + m) * bw + (x / 8 + n));
        if (b[1] != kClpfSkipMark) { for (int k = 0; k < 4; k++) sum[k] += (int)b[k]; filtered = 1; }
      }
    for (int i = 0; i < (luma ? 4 : 1); i++)
      for (int k = 0; k < 4; k++) res[i][k] += sum[k];
    return filtered;
  }
};

// clpf_test_frame (encode_frame.c:597-635) for one plane
static inline void clpf_test_plane(const uint32_t* st, int width, int height, bool luma, double lambda, int max_strength,
                                   int* best_strength, int* best_bs) {
  int64_t sums[4][4];
  memset(sums, 0, sizeof(sums));
  ClpfHost H{st, width / 8};
  const int bs = 8, fb = 7;
  if (!luma) H.rdo(0, 0, fb, width / bs, height / bs, sums, false);
  else
    for (int k = 0; k < (height + (1 << fb) - bs) >> fb; k++)
      for (int l = 0; l < (width + (1 << fb) - bs) >> fb; l++) {
        int h = (height < ((k + 1) << fb) ? height : ((k + 1) << fb)) & ((1 << fb) - 1);
        int w = (width < ((l + 1) << fb) ? width : ((l + 1) << fb)) & ((1 << fb) - 1);
        h += !h << fb;
        w += !w << fb;
        H.rdo(k << fb, l << fb, fb, w / bs, h / bs, sums, true);
      }
  for (int j = 0; j < 4; j++) {
    const int cost = (int)((lambda * (double)sums[j][0] + 6));
    if (!luma) { sums[j][2] += sums[j][2] >> 7; sums[j][3] += sums[j][3] >> 7; }
    for (int i = 0; i < 4; i++) {
      const int i_max = max_strength < 3 ? max_strength : 3;
      if (i > i_max) sums[j][i] = 1 << 30;
      sums[j][i] = ((sums[j][i] + (i && j) * cost) << 4) + j * 4 + i;
    }
  }
  int64_t best = (int64_t)1 << 62;
  for (int j = 0; j < (luma ? 4 : 1); j++)
    for (int i = 0; i < 4; i++)
      if ((i || !j) && sums[j][i] < best) best = sums[j][i];
  best &= 15;
  if (best_bs) *best_bs = (best > 3) * (5 + (best < 12) + (best < 8));
  *best_strength = best ? 1 << ((best - 1) & 3) : 0;
}

// Frame-level CLPF decisions + signalling (encode_frame.c:785-817 with clpf_frame's filter-block loop and
// clpf_decision).  stats: [nby + 2 nbc][4] from pass 1; cells: host copy not needed - skip blocks are marked.
static inline ClpfPlan clpf_plan(const uint32_t* stats, int width, int height, int qp, double lambda, int max_strength) {
  ClpfPlan P;
  if (qp <= 16) { P.bits.push_back({2, 0u}); return P; }
  const int nby = (width / 8) * (height / 8), nbc = (width / 16) * (height / 16);
  int fb_log2 = 0, sy = 0, su = 0, sv = 0;
  clpf_test_plane(stats, width, height, true, lambda, max_strength, &sy, &fb_log2);
  clpf_test_plane(stats + 4 * (size_t)nby, width / 2, height / 2, false, lambda, max_strength, &su, nullptr);
  clpf_test_plane(stats + 4 * (size_t)(nby + nbc), width / 2, height / 2, false, lambda, max_strength, &sv, nullptr);
  int enable_fb = 1;
  if (!fb_log2) { enable_fb = 0; fb_log2 = 7; }
  P.strength[0] = sy; P.strength[1] = su; P.strength[2] = sv; P.fb_log2 = fb_log2; P.enable_fb = enable_fb;
  P.bits.push_back({2, (uint32_t)(sy - (sy == 4))});
  P.bits.push_back({2, (uint32_t)(su - (su == 4))});
  P.bits.push_back({2, (uint32_t)(sv - (sv == 4))});
  const int nfb_h = (width + (1 << fb_log2) - 1) >> fb_log2, nfb_v = (height + (1 << fb_log2) - 1) >> fb_log2;
  P.fb_on.assign((size_t)nfb_h * nfb_v, 0);
  if (sy) {
    P.bits.push_back({2, (uint32_t)((fb_log2 - 4) * enable_fb)});
    const int bw = width / 8, k = sy == 1 ? 1 : (sy == 2 ? 2 : 3);
    for (int fk = 0; fk < nfb_v; fk++)
      for (int fl = 0; fl < nfb_h; fl++) {
        const int x0 = fl << fb_log2, y0 = fk << fb_log2;
        int allskip = 1;
        int64_t sum0 = 0, sum1 = 0;  // reference uses int; values are far below 2^31
        for (int m = 0; m < (1 << fb_log2) / 8; m++)
          for (int n = 0; n < (1 << fb_log2) / 8; n++) {
            const int x = x0 + n * 8, y = y0 + m * 8;
            if (x < width && y < height) {
              const uint32_t* b = stats + 4 * (size_t)((y / 8) * bw + x / 8);
              if (b[1] != kClpfSkipMark) { allskip = 0; sum0 += b[0]; sum1 += b[k]; }
            }
          }
        if (allskip) continue;
        int on = 1;
        if (enable_fb) { on = sum1 < sum0; P.bits.push_back({1, (uint32_t)on}); }
        P.fb_on[(size_t)fk * nfb_h + fl] = (uint8_t)on;
      }
  }
  return P;
}

}  // namespace tk
