// tk_interp_dev.h - temporally interpolated reference frame (interp_ref = 1) as work-item / team code for the GPU
// (SURVEY.md 8f row 3).  Restates common/temporal_interp.c:28-993 with the SIMD variants the oracle binary executes
// (luma-only pyramid, rounded block average); the structure follows the reference function by function:
//   down2x2            scale_frame_down2x2 (:143-160, common_kernels.c:1847) + pad_yuv_frame, any pixel independently
//   estimate_row       motion_estimate_bi (:792-839): make_skip_vector, skip_test, get_cands, adaptive_search_v2
//   merge_block        the merge pass of motion_estimate_bi (:841-861; get_merge_cands, merge_candidate_search)
//   upscale_item       upscale_mv_data_2x2 (:162-184)
//   mot_comp_block     interpolate_frame / mot_comp_avg (:319-375, :865-907)
//   pad_item           pad_yuv_frame of the interpolated picture (enc/mainenc.c:354)
// The block search is raster-serial in the reference: block (i,j) reads the vectors of (i,j-1), (i-1,j-1), (i-1,j)
// and (i-1,j+1).  On the device every 16x16-block ROW is one wavefront that runs two blocks behind the row above
// (progress counters in global memory, agent-scope release/acquire), the lanes share the pixels of each SAD; all other
// phases are flat parallel loops.  The host simulation runs the same code with 1-lane teams, rows in order.
#pragma once
#include "tk_common.h"

namespace tk {
namespace idev {

struct imv {
  int16_t x, y;
};
enum { kBlockStep = 16, kMaxCands = 20, kCostMax = 0x3fffffff, kMaxLevels = 4, kAccBits = 3, kAccRound = 4,
       kLambda = (3000 * kBlockStep) / 16, kLambdaShift = 4, kSkipThr = 8, kBs = kBlockStep / 2, kBbs = kBlockStep, kStep = 2 };

TK_HD int scale_val(int v, int numer, int denom) {  // :60-69
  if (denom == 0) return 0;
  int prod = v * numer;
  if (denom < 0) { denom = -denom; prod = -prod; }
  return prod >= 0 ? (prod + denom / 2) / denom : -((-prod + denom / 2) / denom);
}
TK_HD imv mk(int x, int y) { imv m; m.x = (int16_t)x; m.y = (int16_t)y; return m; }
TK_HD imv scale_mv(imv m, int numer, int denom) {  // :71-85
  if (numer == denom) return m;
  if (numer == -denom) return mk(-m.x, -m.y);
  return mk(scale_val(m.x, numer, denom), scale_val(m.y, numer, denom));
}
TK_HD int iabs_(int a) { return a < 0 ? -a : a; }
TK_HD int clamp_(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// One pyramid level of one stream.
template <typename PIX> struct Level {
  const PIX* pic[2];  // luma at (0,0); pic[0] = reversed ? in1 : in0 (the search's picture order)
  int s[2];
  int w, h, pad;      // picture size, padding of the planes (160 at level 0, 32 above)
  imv* mv[2];         // bw*bh + bw + 2 entries each (raster search result)
  imv* nmv[2];        // merge-pass output = the level's final vectors (read by upscale / motion compensation)
  imv* gmv1;          // guide: final vectors of the coarser level, upscaled to this level's grid (written by upscale_item)
  const imv* guide_mv1;  // = gmv1, or nullptr at the top level
  int guide_reversed, guide_wt0;
  int wt[2], reversed, bw, bh;
  int* prog;          // [bh / kStep]: blocks finished in each block row
  int* ticket;        // row dispenser (device)
};
template <typename PIX> struct Job {
  int levels;
  Level<PIX> lv[kMaxLevels];
  // pyramid planes written by down2x2: level l (1..levels-1) of both inputs, own padding of 32
  PIX* dpic[2][kMaxLevels];
  int dstride[kMaxLevels];
  // level-0 inputs in FRAME order (ref0, ref1) and the output picture (all padded by kPadY)
  Plane3<PIX> ref[2], out;
  int width, height;
};

// ---- 2x2 down-scaling with replicated border: output sample (i,j), i in [-32, h+32), j in [-32, w+32) ------------
template <typename PIX> TK_DEV void down2x2_item(const PIX* in, int in_s, PIX* out, int out_s, int ow, int oh, int i, int j) {
  const int ci = clamp_(i, 0, oh - 1), cj = clamp_(j, 0, ow - 1);
  const PIX* a = in + (ptrdiff_t)(2 * ci) * in_s + 2 * cj;
  const PIX* b = a + in_s;
  out[(ptrdiff_t)i * out_s + j] = (PIX)((((a[0] + b[0] + 1) >> 1) + ((a[1] + b[1] + 1) >> 1)) >> 1);
}

// ---- SAD between the two displaced blocks, coordinates clamped into the padded picture (sad_cost :377-452; the
// in-bounds fast path of the reference is the same sum) ---------------------------------------------------------
template <typename PIX>
TK_DEV uint32_t sad_cost(const Team t, const Level<PIX>& L, int xstart, int ystart, imv mv0, imv mv1, int size, uint32_t cost_start) {
  const int xs0 = xstart + ((mv0.x + kAccRound) >> kAccBits), xs1 = xstart + ((mv1.x + kAccRound) >> kAccBits);
  const int ys0 = ystart + ((mv0.y + kAccRound) >> kAccBits), ys1 = ystart + ((mv1.y + kAccRound) >> kAccBits);
  const int wP = L.w + L.pad, hP = L.h + L.pad, lo = -L.pad;
  const int sh = size == 16 ? 4 : 3;
  int local = 0;
  for (int k = t.rank; k < size * size; k += t.size) {
    const int i = k >> sh, j = k & (size - 1);
    const int x0 = clamp_(j + xs0, lo, wP - 1), x1 = clamp_(j + xs1, lo, wP - 1);
    const int y0 = clamp_(i + ys0, lo, hP - 1), y1 = clamp_(i + ys1, lo, hP - 1);
    local += iabs_((int)L.pic[1][(ptrdiff_t)y1 * L.s[1] + x1] - (int)L.pic[0][(ptrdiff_t)y0 * L.s[0] + x0]);
  }
  return cost_start + (uint32_t)team_sum(t, local);
}

template <typename PIX> TK_DEV int get_mv_cost(const Level<PIX>& L, imv m, int xp, int yp, int lambda) {  // :297-317 (idx 1)
  const int bw = L.bw, xs = kStep, ys = kStep;
  const imv* a = L.mv[1];
  auto d = [&](int pos) { return iabs_(m.x - a[pos].x) + iabs_(m.y - a[pos].y); };
  int diff = 0;
  if (xp == 0 && yp == 0) diff = 0;
  else if (yp > 0 && xp > 0 && xp < bw - xs) diff = d((yp - ys) * bw + xp + xs) + d((yp - ys) * bw + xp) + d((yp - ys) * bw + xp - xs) + d(yp * bw + xp - xs);
  else if (yp == 0) diff = d(xp - xs);
  else if (xp == 0) diff = d((yp - ys) * bw + xp + xs) + d((yp - ys) * bw + xp);
  return (diff * lambda) >> (kLambdaShift + kAccBits);
}

TK_DEV int add_cand(imv* list, int len, imv c) {  // :186-198
  if (len < kMaxCands) {
    list[len] = c;
    for (int i = 0; i < len; i++)
      if (list[i].x == c.x && list[i].y == c.y) return len;
    return len + 1;
  }
  return len;
}
TK_DEV imv absdist_filter(const imv* l, int num) {  // mv_absdist_filter (:701-721): last minimum wins (<=)
  int best = 0, best_cost = kCostMax;
  for (int j = 0; j < num; j++) {
    int cost = 0;
    for (int i = 0; i < num; i++) cost += iabs_(l[i].x - l[j].x) + iabs_(l[i].y - l[j].y);
    if (cost <= best_cost) { best = j; best_cost = cost; }
  }
  return l[best];
}

// One 16x16 block of the raster search: skip vector, skip test, candidates, adaptive search, 2x2 fill.
// Only lane 0 writes the vector fields; every lane computes the same scalars.
template <typename PIX> TK_DEV void estimate_block(const Team t, const Level<PIX>& L, int i, int j) {
  const int bw = L.bw;
  const int pos = i * bw + j;
  // make_skip_vector (:764-776)
  imv skip_mv = mk(0, 0);
  {
    imv vl[3];
    int num = 0;
    if (i > 0 && j < bw - kStep) vl[num++] = L.mv[1][(i - kStep) * bw + j + kStep];
    if (j > 0) vl[num++] = L.mv[1][i * bw + j - kStep];
    if (i > 0) vl[num++] = L.mv[1][(i - kStep) * bw + j];
    if (num) skip_mv = absdist_filter(vl, num);
  }
  const imv scaled_skip = scale_mv(skip_mv, -L.wt[1], L.wt[0]);
  // skip_test (:455-574, chroma branch compiled out): four 8x8 SADs, all fully inside the padded picture
  const int xstart = j * kBs, ystart = i * kBs;
  int skip = 1;
  {
    const int wP = L.w + L.pad, hP = L.h + L.pad, lo = -L.pad;
    for (int p = ystart; p < ystart + kBbs && skip; p += 8)
      for (int q = xstart; q < xstart + kBbs && skip; q += 8) {
        const int x0 = q + ((scaled_skip.x + kAccRound) >> kAccBits), x1 = q + ((skip_mv.x + kAccRound) >> kAccBits);
        const int y0 = p + ((scaled_skip.y + kAccRound) >> kAccBits), y1 = p + ((skip_mv.y + kAccRound) >> kAccBits);
        if (x0 >= lo && x0 + 8 <= wP && y0 >= lo && y0 + 8 <= hP && x1 >= lo && x1 + 8 <= wP && y1 >= lo && y1 + 8 <= hP) {
          int local = 0;
          for (int k = t.rank; k < 64; k += t.size) {
            const int a = k >> 3, b = k & 7;
            local += iabs_((int)L.pic[0][(ptrdiff_t)(y0 + a) * L.s[0] + x0 + b] - (int)L.pic[1][(ptrdiff_t)(y1 + a) * L.s[1] + x1 + b]);
          }
          if (team_sum(t, local) > kSkipThr * 8 * 8) skip = 0;
        } else skip = 0;
      }
  }
  imv m0 = L.mv[0][pos], m1 = L.mv[1][pos];
  if (skip) { m1 = skip_mv; m0 = scaled_skip; }
  if (!skip) {
    // get_cands (:219-269)
    imv cl[kMaxCands];
    int len = add_cand(cl, 0, mk(0, 0));
    const int guided = L.guide_mv1 != nullptr;
    if (guided) {
      const int numer = (L.reversed == L.guide_reversed) ? L.wt[0] : -L.wt[0];
      len = add_cand(cl, len, scale_mv(L.guide_mv1[pos], numer, L.guide_wt0));
    }
    if (i > 0 && j < bw - kStep) len = add_cand(cl, len, L.mv[1][(i - kStep) * bw + j + kStep]);
    if (j > 0) len = add_cand(cl, len, L.mv[1][i * bw + j - kStep]);
    if (i > 0) len = add_cand(cl, len, L.mv[1][(i - kStep) * bw + j]);
    // adaptive_search_v2 (:577-659)
    imv best_mv = cl[0], best_scaled = scale_mv(best_mv, -L.wt[1], L.wt[0]);
    uint32_t best_cost = kCostMax;
    const int lambda = guided ? kLambda / 4 : kLambda;
    for (int c = 0; c < len; c++) {
      imv r1 = cl[c], r0 = scale_mv(cl[c], -L.wt[1], L.wt[0]);
      uint32_t cost = sad_cost(t, L, xstart, ystart, r0, r1, kBbs, (uint32_t)get_mv_cost(L, cl[c], j, i, lambda));
      if ((((4 + c) * cost) / 8) < best_cost) {
        int shift = guided ? 0 + kAccBits : 3 + kAccBits;
        int count = guided ? 8 : 64;
        while (shift >= kAccBits && count > 0) {
          const int off = 1 << shift;
          const imv ctr = r1;
          int better = 0;
          for (int q = 0; q < 4; q++) {
            const imv rmv = q == 0 ? mk(ctr.x - off, ctr.y) : q == 1 ? mk(ctr.x + off, ctr.y) : q == 2 ? mk(ctr.x, ctr.y - off) : mk(ctr.x, ctr.y + off);
            const imv rs = scale_mv(rmv, -L.wt[1], L.wt[0]);
            const uint32_t bc = sad_cost(t, L, xstart, ystart, rs, rmv, kBbs, (uint32_t)get_mv_cost(L, rmv, j, i, lambda));
            if (bc < cost) { cost = bc; r1 = rmv; r0 = rs; better = 1; }
          }
          if (!better) shift--;
          count -= 4;
        }
      }
      if (cost < best_cost) { best_mv = r1; best_scaled = r0; best_cost = cost; }
    }
    m1 = best_mv;
    m0 = best_scaled;
  }
  t.sync();
  if (t.rank == 0)
    for (int q = 0; q < kStep; q++)
      for (int p = 0; p < kStep; p++) { L.mv[0][pos + q * bw + p] = m0; L.mv[1][pos + q * bw + p] = m1; }
  t.sync();
}

// merge pass, one 8x8 unit (i, j) in [0,bh) x [0,bw): reads mv[1], writes nmv[0/1]
template <typename PIX> TK_DEV void merge_block(const Team t, const Level<PIX>& L, int i, int j) {
  const int bw = L.bw, bh = L.bh;
  const int off = (i & 1) ? 2 : 1;  // sic: the row parity selects both offsets
  imv cl[kMaxCands];
  int len = add_cand(cl, 0, L.mv[1][i * bw + j]);
  if (i - off >= 0) len = add_cand(cl, len, L.mv[1][(i - off) * bw + j]);
  if (i + off < bh) len = add_cand(cl, len, L.mv[1][(i + off) * bw + j]);
  if (j - off >= 0) len = add_cand(cl, len, L.mv[1][i * bw + j - off]);
  if (j + off < bw) len = add_cand(cl, len, L.mv[1][i * bw + j + off]);
  imv o0 = L.mv[0][i * bw + j], o1 = L.mv[1][i * bw + j];
  if (len > 1) {
    uint32_t best = kCostMax;
    imv bm = mk(0, 0), bs = mk(0, 0);
    for (int c = 0; c < len; c++) {
      const imv s0 = scale_mv(cl[c], -L.wt[1], L.wt[0]);
      const uint32_t bc = sad_cost(t, L, j * kBs, i * kBs, s0, cl[c], kBs, 0u);
      if (bc < best) { best = bc; bm = cl[c]; bs = s0; }
    }
    o1 = bm;
    o0 = bs;
  }
  if (t.rank == 0) { L.nmv[0][i * bw + j] = o0; L.nmv[1][i * bw + j] = o1; }
}

// upscale_mv_data_2x2: entry (i, j) of the finer level's guide from the coarser level's vectors (:162-184)
TK_DEV void upscale_item(const imv* in_mv1, int in_bw, imv* out_mv1, int out_bw, int i, int j) {
  const imv v = in_mv1[(i / 2) * in_bw + (j / 2)];
  out_mv1[i * out_bw + j] = mk((int16_t)(v.x << 1), (int16_t)(v.y << 1));
}

// mot_comp_avg (:319-375) of one block of one component; lanes share the samples
template <typename PIX>
TK_DEV void mot_comp_block(const Team t, int xstart, int ystart, const PIX* ref0, int s0, const PIX* ref1, int s1, PIX* pic, int sp, imv mv0,
                           imv mv1, int wP, int hP, int pad, int size) {
  const int xs0 = xstart + ((mv0.x + kAccRound) >> kAccBits), xs1 = xstart + ((mv1.x + kAccRound) >> kAccBits);
  const int ys0 = ystart + ((mv0.y + kAccRound) >> kAccBits), ys1 = ystart + ((mv1.y + kAccRound) >> kAccBits);
  PIX* p = pic + (ptrdiff_t)ystart * sp + xstart;
  const int in0 = xs0 >= -pad && xs0 + size <= wP && ys0 >= -pad && ys0 + size <= hP;
  const int in1 = xs1 >= -pad && xs1 + size <= wP && ys1 >= -pad && ys1 + size <= hP;
  const int sh = size == 8 ? 3 : 2;
  for (int k = t.rank; k < size * size; k += t.size) {
    const int i = k >> sh, j = k & (size - 1);
    int v;
    if (in0 && in1) v = ((int)ref0[(ptrdiff_t)(ys0 + i) * s0 + xs0 + j] + (int)ref1[(ptrdiff_t)(ys1 + i) * s1 + xs1 + j] + 1) / 2;
    else if (in1) v = ref1[(ptrdiff_t)(ys1 + i) * s1 + xs1 + j];
    else if (in0) v = ref0[(ptrdiff_t)ys0 * s0 + xs0 + (ptrdiff_t)i * s1 + j];  // sic: row pitch s1 (:356)
    else {
      const int x0 = clamp_(j + xs0, -pad, wP - 1), x1 = clamp_(j + xs1, -pad, wP - 1);
      const int y0 = clamp_(i + ys0, -pad, hP - 1), y1 = clamp_(i + ys1, -pad, hP - 1);
      v = ((int)ref0[(ptrdiff_t)y0 * s0 + x0] + (int)ref1[(ptrdiff_t)y1 * s1 + x1] + 1) / 2;
    }
    p[i * sp + j] = (PIX)v;
  }
}

// interpolate_frame (:865-907) for 8x8 unit (yp, xp) of the level-0 grid: luma 8x8 + two chroma 4x4
template <typename PIX> TK_DEV void mot_comp_unit(const Team t, const Job<PIX>& J, int yp, int xp) {
  const Level<PIX>& D = J.lv[0];
  const Plane3<PIX>& p0 = D.reversed ? J.ref[1] : J.ref[0];
  const Plane3<PIX>& p1 = D.reversed ? J.ref[0] : J.ref[1];
  const int pad = kBs / 2, wP = J.width + pad, hP = J.height + pad;
  const imv m0 = D.nmv[0][yp * D.bw + xp], m1 = D.nmv[1][yp * D.bw + xp];
  mot_comp_block(t, xp * kBs, yp * kBs, p0.y, p0.sy, p1.y, p1.sy, J.out.y, J.out.sy, m0, m1, wP, hP, pad, kBs);
  const imv c1 = mk((int16_t)(m1.x >> 1), (int16_t)(m1.y >> 1));
  const imv c0 = scale_mv(c1, -D.wt[1], D.wt[0]);
  const int bs = kBs / 2;
  mot_comp_block(t, xp * bs, yp * bs, p0.u, p0.sc, p1.u, p1.sc, J.out.u, J.out.sc, c0, c1, wP >> 1, hP >> 1, pad >> 1, bs);
  mot_comp_block(t, xp * bs, yp * bs, p0.v, p0.sc, p1.v, p1.sc, J.out.v, J.out.sc, c0, c1, wP >> 1, hP >> 1, pad >> 1, bs);
}

// pad_yuv_frame of the output: item = one padded row of one plane (it < total rows); lanes stride along the row and
// write ONLY border samples, each the nearest picture sample (reads stay inside the picture: no ordering needed)
template <typename PIX> TK_DEV void pad_item(const Job<PIX>& J, int it, int lane, int nlanes) {
  const int py = kPadY, pc = kPadY / 2;
  const int hy = J.height + 2 * py, hc = J.height / 2 + 2 * pc;
  PIX* pl;
  int w, h, pad, s, row;
  if (it < hy) { row = it - py; pl = J.out.y; w = J.width; h = J.height; pad = py; s = J.out.sy; }
  else if (it < hy + hc) { row = it - hy - pc; pl = J.out.u; w = J.width / 2; h = J.height / 2; pad = pc; s = J.out.sc; }
  else { row = it - hy - hc - pc; pl = J.out.v; w = J.width / 2; h = J.height / 2; pad = pc; s = J.out.sc; }
  const int sr = clamp_(row, 0, h - 1);
  const PIX* src = pl + (ptrdiff_t)sr * s;
  PIX* dst = pl + (ptrdiff_t)row * s;
  const int inside = row >= 0 && row < h;
  for (int x = -pad + lane; x < w + pad; x += nlanes)
    if (!inside || x < 0 || x >= w) dst[x] = src[clamp_(x, 0, w - 1)];
}

}  // namespace idev
}  // namespace tk
