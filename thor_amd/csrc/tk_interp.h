// tk_interp.h - HOST code: temporally interpolated reference frame for hierarchical-B coding
// (interp_ref = 1).  The reference runs this on the CPU between frames
// (enc/mainenc.c:353, dec/decode_frame.c:110); it is normative (the decoder repeats it), so it is
// restated here exactly: 2x2 luma pyramid (up to 4 levels), raster-order bidirectional block motion
// search with neighbour / coarser-level candidates and a cross refinement, a merge pass on 8x8 units,
// and rounded-average motion compensation at level 0.
// Specification followed: common/temporal_interp.c:28-993 (every function cited below) with the SIMD
// variants the oracle binary executes (scale_frame_down2x2_simd common_kernels.c:1847 - luma only,
// block_avg_simd :38, sad_calc_simd_unaligned :68 - all numerically equal to the scalar code).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace tk {
namespace interp {

struct imv {
  int16_t x, y;
};

template <typename PIX> struct HFrame {  // padded planar frame on the host (layout of create_yuv_frame, common_frame.c:435)
  std::vector<PIX> by, bc;
  PIX *y = nullptr, *u = nullptr, *v = nullptr;
  int w = 0, h = 0, sy = 0, sc = 0, pad = 0;
  void alloc(int w_, int h_, int pad_) {
    w = w_; h = h_; pad = pad_;
    sy = (w + 2 * pad + 15) & ~15;
    sc = (w / 2 + 2 * (pad / 2) + 15) & ~15;
    by.assign((size_t)(h + 2 * pad) * sy + 64, 0);
    size_t ac = (size_t)(h / 2 + 2 * (pad / 2)) * sc + 64;
    bc.assign(2 * ac, 0);
    y = by.data() + (size_t)pad * sy + pad;
    u = bc.data() + (size_t)(pad / 2) * sc + pad / 2;
    v = u + ac;
  }
  static void pad_plane(PIX* p, int w, int h, int s, int pad) {  // pad_yuv_frame (common_frame.c:657-735)
    for (int i = 0; i < h; i++) {
      PIX* r = p + (size_t)i * s;
      for (int j = 1; j <= pad; j++) { r[-j] = r[0]; r[w - 1 + j] = r[w - 1]; }
    }
    for (int i = 1; i <= pad; i++) {
      memcpy(p - (ptrdiff_t)i * s - pad, p - pad, (size_t)(w + 2 * pad) * sizeof(PIX));
      memcpy(p + (ptrdiff_t)(h - 1 + i) * s - pad, p + (ptrdiff_t)(h - 1) * s - pad, (size_t)(w + 2 * pad) * sizeof(PIX));
    }
  }
  void pad_luma() { pad_plane(y, w, h, sy, pad); }
  void pad_all() { pad_plane(y, w, h, sy, pad); pad_plane(u, w / 2, h / 2, sc, pad / 2); pad_plane(v, w / 2, h / 2, sc, pad / 2); }
};

enum { kBlockStep = 16, kMaxCands = 20, kCostMax = 0x3fffffff, kMaxLevels = 4, kAccBits = 3, kAccRound = 4, kLambda = (3000 * kBlockStep) / 16, kLambdaShift = 4, kSkipThr = 8 };

struct MvData {  // mv_data_t (temporal_interp.c:38-58, alloc_mv_data :87-134, interpolating case)
  std::vector<imv> mv[2];
  std::vector<int> bgmap;
  int wt[2], reversed, bbs, bs, step, bw, bh;
  imv skip_mv, scaled_skip_mv;
  void init(int w, int h, int ratio, int k) {
    bs = kBlockStep / 2; bbs = kBlockStep; step = bbs / bs;
    bw = step * ((w + (bbs - 1)) / bbs);
    bh = step * ((h + (bbs - 1)) / bbs);
    mv[0].assign((size_t)bw * bh + bw + 2, imv{0, 0});
    mv[1].assign((size_t)bw * bh + bw + 2, imv{0, 0});
    bgmap.assign((size_t)bw * bh + bw + 2, 0);
    reversed = k > ratio / 2;
    wt[0] = reversed ? k : ratio - k;
    wt[1] = ratio - wt[0];
    skip_mv = scaled_skip_mv = imv{0, 0};
  }
};

static inline int scale_val(int v, int numer, int denom) {  // :60-69
  if (denom == 0) return 0;
  int prod = v * numer;
  if (denom < 0) { denom = -denom; prod = -prod; }
  return prod >= 0 ? (prod + denom / 2) / denom : -((-prod + denom / 2) / denom);
}
static inline imv scale_mv(imv m, int numer, int denom) {  // :71-85
  if (numer == denom) return m;
  if (numer == -denom) return imv{(int16_t)-m.x, (int16_t)-m.y};
  return imv{(int16_t)scale_val(m.x, numer, denom), (int16_t)scale_val(m.y, numer, denom)};
}

template <typename PIX> static void down2x2(const HFrame<PIX>& in, HFrame<PIX>& out) {  // luma only (SIMD variant)
  for (int i = 0; i < out.h; i++)
    for (int j = 0; j < out.w; j++) {
      const PIX* a = in.y + (size_t)(2 * i) * in.sy + 2 * j;
      const PIX* b = a + in.sy;
      out.y[(size_t)i * out.sy + j] = (PIX)((((a[0] + b[0] + 1) >> 1) + ((a[1] + b[1] + 1) >> 1)) >> 1);
    }
  out.pad_luma();
}

static inline int add_cand(imv* list, int max, int len, imv c) {  // :186-198
  if (len < max) {
    list[len] = c;
    for (int i = 0; i < len; i++)
      if (list[i].x == c.x && list[i].y == c.y) return len;
    return len + 1;
  }
  return len;
}

template <typename PIX> struct Pair {
  const HFrame<PIX>* pic[2];
};

// sad_cost (:377-452), luma only
template <typename PIX> static uint32_t sad_cost(int xstart, int ystart, const Pair<PIX>& P, const imv mv[2], int size, uint32_t cost_start) {
  int xs[2], ys[2];
  xs[0] = xstart + ((mv[0].x + kAccRound) >> kAccBits); xs[1] = xstart + ((mv[1].x + kAccRound) >> kAccBits);
  ys[0] = ystart + ((mv[0].y + kAccRound) >> kAccBits); ys[1] = ystart + ((mv[1].y + kAccRound) >> kAccBits);
  const int pady = P.pic[0]->pad, wP = P.pic[0]->w + pady, hP = P.pic[0]->h + pady;
  const int s0 = P.pic[0]->sy, s1 = P.pic[1]->sy;
  uint32_t c = cost_start;
  if (xs[0] >= -pady && xs[0] + size <= wP && ys[0] >= -pady && ys[0] + size <= hP && xs[1] >= -pady && xs[1] + size <= wP &&
      ys[1] >= -pady && ys[1] + size <= hP) {
    const PIX* p0 = P.pic[0]->y + (ptrdiff_t)ys[0] * s0 + xs[0];
    const PIX* p1 = P.pic[1]->y + (ptrdiff_t)ys[1] * s1 + xs[1];
    for (int i = 0; i < size; i++)
      for (int j = 0; j < size; j++) c += (uint32_t)abs((int)p1[i * s1 + j] - (int)p0[i * s0 + j]);
  } else {
    const PIX* p0 = P.pic[0]->y;
    const PIX* p1 = P.pic[1]->y;
    auto cl = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    for (int i = 0; i < size; i++)
      for (int j = 0; j < size; j++) {
        int x0 = cl(j + xs[0], -pady, wP - 1), x1 = cl(j + xs[1], -pady, wP - 1);
        int y0 = cl(i + ys[0], -pady, hP - 1), y1 = cl(i + ys[1], -pady, hP - 1);
        c += (uint32_t)abs((int)p1[(ptrdiff_t)y1 * s1 + x1] - (int)p0[(ptrdiff_t)y0 * s0 + x0]);
      }
  }
  return c;
}

static inline int get_mv_cost(imv m, const MvData& D, int xp, int yp, int xs, int ys, int lambda) {  // :297-317 (idx = 1)
  int diff = 0;
  const int bw = D.bw;
  const imv* a = D.mv[1].data();
  auto d = [&](int pos) { return abs(m.x - a[pos].x) + abs(m.y - a[pos].y); };
  if (xp == 0 && yp == 0) diff = 0;
  else if (yp > 0 && xp > 0 && xp < D.bw - xs) diff = d((yp - ys) * bw + xp + xs) + d((yp - ys) * bw + xp) + d((yp - ys) * bw + xp - xs) + d(yp * bw + xp - xs);
  else if (yp == 0) diff = d(xp - xs);
  else if (xp == 0) diff = d((yp - ys) * bw + xp + xs) + d((yp - ys) * bw + xp);
  return (diff * lambda) >> (kLambdaShift + kAccBits);
}

static inline imv absdist_filter(const imv* l, int num) {  // mv_absdist_filter (:701-721): last minimum wins (<=)
  int best = 0, best_cost = kCostMax;
  for (int j = 0; j < num; j++) {
    int cost = 0;
    for (int i = 0; i < num; i++) cost += abs(l[i].x - l[j].x) + abs(l[i].y - l[j].y);
    if (cost <= best_cost) { best = j; best_cost = cost; }
  }
  return l[best];
}

// skip_test (:455-574, chroma branch compiled out)
template <typename PIX> static void skip_test(MvData& D, const Pair<PIX>& P, int xp, int yp) {
  const int xstart = xp * D.bs, ystart = yp * D.bs, pos = yp * D.bw + xp, size = D.bbs;
  const imv mv1 = D.skip_mv, mv0 = D.scaled_skip_mv;
  const int thr = kSkipThr * 8 * 8;
  int skip = 1;
  const int pady = P.pic[0]->pad, padx = P.pic[0]->pad, hP = P.pic[0]->h + pady, wP = P.pic[0]->w + padx;
  const int s0 = P.pic[0]->sy, s1 = P.pic[1]->sy;
  for (int p = ystart; p < ystart + size && skip; p += 8)
    for (int q = xstart; q < xstart + size && skip; q += 8) {
      int x0 = q + ((mv0.x + kAccRound) >> kAccBits), x1 = q + ((mv1.x + kAccRound) >> kAccBits);
      int y0 = p + ((mv0.y + kAccRound) >> kAccBits), y1 = p + ((mv1.y + kAccRound) >> kAccBits);
      if (x0 >= -padx && x0 + 8 <= wP && y0 >= -pady && y0 + 8 <= hP && x1 >= -padx && x1 + 8 <= wP && y1 >= -pady && y1 + 8 <= hP) {
        int sum = 0;
        const PIX* r0 = P.pic[0]->y + (ptrdiff_t)y0 * s0 + x0;
        const PIX* r1 = P.pic[1]->y + (ptrdiff_t)y1 * s1 + x1;
        for (int i = 0; i < 8; i++)
          for (int j = 0; j < 8; j++) sum += abs((int)r0[i * s0 + j] - (int)r1[i * s1 + j]);
        if (sum > thr) { skip = 0; break; }
      } else { skip = 0; break; }
    }
  if (skip) { D.bgmap[pos] = 1; D.mv[1][pos] = D.skip_mv; D.mv[0][pos] = D.scaled_skip_mv; }
  const int bw = D.bw;
  for (int k = 0; k < 2; k++) { D.mv[k][pos + 1] = D.mv[k][pos]; D.mv[k][pos + bw] = D.mv[k][pos]; D.mv[k][pos + bw + 1] = D.mv[k][pos]; }
  D.bgmap[pos + 1] = D.bgmap[pos]; D.bgmap[pos + bw] = D.bgmap[pos]; D.bgmap[pos + bw + 1] = D.bgmap[pos];
}

// adaptive_search_v2 (:577-659)
template <typename PIX> static void adaptive_search(MvData& D, int guided, const imv* cand, int ncand, const Pair<PIX>& P, int xp, int yp, int xstep, int ystep) {
  const int xstart = xp * D.bs, ystart = yp * D.bs, size = D.bbs;
  imv best_mv = cand[0], best_scaled = scale_mv(best_mv, -D.wt[1], D.wt[0]);
  uint32_t best_cost = kCostMax;
  imv refine[kMaxCands], refine_s[kMaxCands];
  uint32_t costs[kMaxCands];
  const int lambda = guided ? kLambda / 4 : kLambda;
  for (int c = 0; c < ncand; c++) {
    imv mv[2];
    mv[1] = cand[c];
    mv[0] = scale_mv(cand[c], -D.wt[1], D.wt[0]);
    costs[c] = (uint32_t)get_mv_cost(cand[c], D, xp, yp, xstep, ystep, lambda);
    costs[c] = sad_cost(xstart, ystart, P, mv, size, costs[c]);
    refine[c] = mv[1];
    refine_s[c] = mv[0];
    if ((((4 + c) * costs[c]) / 8) < best_cost) {
      int shift = guided ? 0 + kAccBits : 3 + kAccBits;
      int count = guided ? 8 : 64;
      while (shift >= kAccBits && count > 0) {
        const int off = 1 << shift;
        const imv ctr = refine[c];
        const imv cross[4] = {{(int16_t)(ctr.x - off), ctr.y}, {(int16_t)(ctr.x + off), ctr.y}, {ctr.x, (int16_t)(ctr.y - off)}, {ctr.x, (int16_t)(ctr.y + off)}};
        int better = 0;
        for (int i = 0; i < 4; i++) {
          imv rmv = cross[i];
          mv[1] = rmv;
          mv[0] = scale_mv(rmv, -D.wt[1], D.wt[0]);
          uint32_t bc = (uint32_t)get_mv_cost(rmv, D, xp, yp, xstep, ystep, lambda);
          bc = sad_cost(xstart, ystart, P, mv, size, bc);
          if (bc < costs[c]) { costs[c] = bc; refine[c] = rmv; refine_s[c] = mv[0]; better = 1; }
        }
        if (!better) shift--;
        count -= 4;
      }
    }
    if (costs[c] < best_cost) { best_mv = refine[c]; best_scaled = refine_s[c]; best_cost = costs[c]; }
  }
  D.mv[1][yp * D.bw + xp] = best_mv;
  D.mv[0][yp * D.bw + xp] = best_scaled;
}

// motion_estimate_bi (:792-863)
template <typename PIX> static void estimate(MvData& D, const MvData* guide, const HFrame<PIX>& in0, const HFrame<PIX>& in1) {
  const int bw = D.bw, bh = D.bh, step = D.step;
  if (!guide) { std::fill(D.mv[0].begin(), D.mv[0].end(), imv{0, 0}); std::fill(D.mv[1].begin(), D.mv[1].end(), imv{0, 0}); }
  std::fill(D.bgmap.begin(), D.bgmap.end(), 0);
  Pair<PIX> P;
  P.pic[0] = D.reversed ? &in1 : &in0;
  P.pic[1] = D.reversed ? &in0 : &in1;
  imv cl[kMaxCands];
  for (int i = 0; i < bh; i += step)
    for (int j = 0; j < bw; j += step) {
      {  // make_skip_vector (:764-776)
        D.skip_mv = imv{0, 0};
        imv vl[3];
        int num = 0;
        if (i > 0 && j < bw - step) vl[num++] = D.mv[1][(i - step) * bw + j + step];
        if (j > 0) vl[num++] = D.mv[1][i * bw + j - step];
        if (i > 0) vl[num++] = D.mv[1][(i - step) * bw + j];
        if (num) D.skip_mv = absdist_filter(vl, num);
        D.scaled_skip_mv = scale_mv(D.skip_mv, -D.wt[1], D.wt[0]);
      }
      skip_test(D, P, j, i);
      const int pos = i * bw + j;
      if (D.bgmap[pos] == 0) {
        // get_cands (:219-269)
        int len = add_cand(cl, kMaxCands, 0, imv{0, 0});
        if (guide) {
          int numer = (D.reversed == guide->reversed) ? D.wt[0] : -D.wt[0];
          len = add_cand(cl, kMaxCands, len, scale_mv(guide->mv[1][pos], numer, guide->wt[0]));
        }
        if (i > 0 && j < bw - step) len = add_cand(cl, kMaxCands, len, D.mv[1][(i - step) * bw + j + step]);
        if (j > 0) len = add_cand(cl, kMaxCands, len, D.mv[1][i * bw + j - step]);
        if (i > 0) len = add_cand(cl, kMaxCands, len, D.mv[1][(i - step) * bw + j]);
        adaptive_search(D, guide != nullptr, cl, len, P, j, i, step, step);
      }
      const imv m0 = D.mv[0][pos], m1 = D.mv[1][pos];
      const int bg = D.bgmap[pos];
      for (int q = 0; q < step; q++)
        for (int p = 0; p < step; p++) { D.mv[0][pos + q * bw + p] = m0; D.mv[1][pos + q * bw + p] = m1; D.bgmap[pos + q * bw + p] = bg; }
    }
  // merge pass on the small blocks (:841-861; get_merge_cands :200-217, merge_candidate_search :661-699)
  std::vector<imv> n0((size_t)bw * bh), n1((size_t)bw * bh);
  for (int i = 0; i < bh; i++)
    for (int j = 0; j < bw; j++) {
      const int off = (i & 1) ? 2 : 1;  // sic: the row parity selects both offsets
      int len = add_cand(cl, kMaxCands, 0, D.mv[1][i * bw + j]);
      if (i - off >= 0) len = add_cand(cl, kMaxCands, len, D.mv[1][(i - off) * bw + j]);
      if (i + off < bh) len = add_cand(cl, kMaxCands, len, D.mv[1][(i + off) * bw + j]);
      if (j - off >= 0) len = add_cand(cl, kMaxCands, len, D.mv[1][i * bw + j - off]);
      if (j + off < bw) len = add_cand(cl, kMaxCands, len, D.mv[1][i * bw + j + off]);
      if (len > 1) {
        uint32_t best = kCostMax;
        imv bm{0, 0}, bs{0, 0};
        for (int c = 0; c < len; c++) {
          imv mv[2];
          mv[1] = cl[c];
          mv[0] = scale_mv(cl[c], -D.wt[1], D.wt[0]);
          uint32_t bc = sad_cost(j * D.bs, i * D.bs, P, mv, D.bs, 0u);
          if (bc < best) { best = bc; bm = cl[c]; bs = mv[0]; }
        }
        n1[(size_t)i * bw + j] = bm;
        n0[(size_t)i * bw + j] = bs;
      } else {
        n0[(size_t)i * bw + j] = D.mv[0][i * bw + j];
        n1[(size_t)i * bw + j] = D.mv[1][i * bw + j];
      }
    }
  std::copy(n0.begin(), n0.end(), D.mv[0].begin());
  std::copy(n1.begin(), n1.end(), D.mv[1].begin());
}

// mot_comp_avg (:319-375)
template <typename PIX>
static void mot_comp_avg(int xstart, int ystart, const PIX* ref0, int s0, const PIX* ref1, int s1, PIX* pic, int sp, imv mv0, imv mv1, int wP, int hP, int pad, int size) {
  int xs[2], ys[2];
  xs[0] = xstart + ((mv0.x + kAccRound) >> kAccBits); xs[1] = xstart + ((mv1.x + kAccRound) >> kAccBits);
  ys[0] = ystart + ((mv0.y + kAccRound) >> kAccBits); ys[1] = ystart + ((mv1.y + kAccRound) >> kAccBits);
  PIX* p = pic + (ptrdiff_t)ystart * sp + xstart;
  const bool in0 = xs[0] >= -pad && xs[0] + size <= wP && ys[0] >= -pad && ys[0] + size <= hP;
  const bool in1 = xs[1] >= -pad && xs[1] + size <= wP && ys[1] >= -pad && ys[1] + size <= hP;
  if (in0 && in1) {
    const PIX* r0 = ref0 + (ptrdiff_t)ys[0] * s0 + xs[0];
    const PIX* r1 = ref1 + (ptrdiff_t)ys[1] * s1 + xs[1];
    for (int i = 0; i < size; i++)
      for (int j = 0; j < size; j++) p[i * sp + j] = (PIX)(((int)r0[i * s0 + j] + (int)r1[i * s1 + j] + 1) / 2);
  } else if (in1) {
    const PIX* r1 = ref1 + (ptrdiff_t)ys[1] * s1 + xs[1];
    for (int i = 0; i < size; i++) memcpy(p + i * sp, r1 + i * s1, size * sizeof(PIX));
  } else if (in0) {
    const PIX* r0 = ref0 + (ptrdiff_t)ys[0] * s0 + xs[0];
    for (int i = 0; i < size; i++) memcpy(p + i * sp, r0 + i * s1, size * sizeof(PIX));  // sic: stride s1 (:356)
  } else {
    auto cl = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    for (int i = 0; i < size; i++)
      for (int j = 0; j < size; j++) {
        int x0 = cl(j + xs[0], -pad, wP - 1), x1 = cl(j + xs[1], -pad, wP - 1);
        int y0 = cl(i + ys[0], -pad, hP - 1), y1 = cl(i + ys[1], -pad, hP - 1);
        p[i * sp + j] = (PIX)(((int)ref0[(ptrdiff_t)y0 * s0 + x0] + (int)ref1[(ptrdiff_t)y1 * s1 + x1] + 1) / 2);
      }
  }
}

// interpolate_frames (:909-993) + interpolate_frame / interpolate_comp (:865-907).
// ref0/ref1: padded (160) reference frames; out: padded frame receiving the interpolated picture
// (the caller pads it afterwards like enc/mainenc.c:354).
template <typename PIX> static void interpolate_frames(HFrame<PIX>& out, const HFrame<PIX>& ref0, const HFrame<PIX>& ref1, int ratio, int pos) {
  const int w = ref0.w, h = ref0.h;
  const int mn = w < h ? w : h;
  int max_levels = (int)(log10((double)mn) / log10(2.0) - 4.0);
  if (max_levels > kMaxLevels) max_levels = kMaxLevels;
  std::vector<MvData> mvd(max_levels), spat(max_levels);
  std::vector<HFrame<PIX>> d0(max_levels), d1(max_levels);
  for (int j = 0; j < max_levels; j++) { mvd[j].init(w >> j, h >> j, ratio, pos); spat[j].init(w >> j, h >> j, ratio, pos); }
  for (int l = 1; l < max_levels; l++) {
    d0[l].alloc(w >> l, h >> l, 32);
    d1[l].alloc(w >> l, h >> l, 32);
    down2x2(l == 1 ? ref0 : d0[l - 1], d0[l]);
    down2x2(l == 1 ? ref1 : d1[l - 1], d1[l]);
  }
  for (int lvl = max_levels - 1; lvl >= 0; --lvl) {
    const MvData* guide = lvl != max_levels - 1 ? &spat[lvl] : nullptr;
    const HFrame<PIX>& a = lvl == 0 ? ref0 : d0[lvl];
    const HFrame<PIX>& b = lvl == 0 ? ref1 : d1[lvl];
    estimate(mvd[lvl], guide, a, b);
    if (lvl == 0) {
      MvData& D = mvd[0];
      const HFrame<PIX>* p0 = D.reversed ? &ref1 : &ref0;
      const HFrame<PIX>* p1 = D.reversed ? &ref0 : &ref1;
      const int pad = D.bs / 2, wP = w + pad, hP = h + pad;
      for (int comp = 0; comp < 3; comp++) {
        const int chroma = comp > 0;
        const int bs = chroma ? D.bs / 2 : D.bs;
        const PIX* q0 = comp == 0 ? p0->y : (comp == 1 ? p0->u : p0->v);
        const PIX* q1 = comp == 0 ? p1->y : (comp == 1 ? p1->u : p1->v);
        PIX* o = comp == 0 ? out.y : (comp == 1 ? out.u : out.v);
        const int s0 = chroma ? p0->sc : p0->sy, s1 = chroma ? p1->sc : p1->sy, so = chroma ? out.sc : out.sy;
        for (int yp = 0; yp < D.bh; yp++)
          for (int xp = 0; xp < D.bw; xp++) {
            imv m0 = D.mv[0][yp * D.bw + xp], m1 = D.mv[1][yp * D.bw + xp];
            if (chroma) { m1.x >>= 1; m1.y >>= 1; m0 = scale_mv(m1, -D.wt[1], D.wt[0]); }
            mot_comp_avg(xp * bs, yp * bs, q0, s0, q1, s1, o, so, m0, m1, chroma ? wP >> 1 : wP, chroma ? hP >> 1 : hP, chroma ? pad >> 1 : pad, bs);
          }
      }
    } else {
      // upscale_mv_data_2x2 (:162-184)
      MvData& in = mvd[lvl];
      MvData& o = spat[lvl - 1];
      std::fill(o.mv[0].begin(), o.mv[0].end(), imv{0, 0});
      std::fill(o.mv[1].begin(), o.mv[1].end(), imv{0, 0});
      for (int i = 0; i < o.bh; i++)
        for (int j = 0; j < o.bw; j++) {
          const int po = i * o.bw + j, pi = (i / 2) * in.bw + (j / 2);
          o.mv[1][po] = imv{(int16_t)(in.mv[1][pi].x << 1), (int16_t)(in.mv[1][pi].y << 1)};
          o.mv[0][po] = scale_mv(o.mv[1][po], -o.wt[1], o.wt[0]);
        }
    }
  }
}

}  // namespace interp
}  // namespace tk
