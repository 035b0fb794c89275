// tk_filters.h - frame-level in-loop filters and reference-frame creation, work-item parallel.
// Specification followed: common/common_frame.c:47-352 (deblock_frame_y, MODIFIED_DEBLOCK_TEST /
// NEW_MV_TEST / NEW_DEBLOCK_FILTER variants as compiled), :354-433 (deblock_frame_uv), :657-763
// (pad_yuv_frame / create_reference_frame).  Every 8-sample edge segment is independent inside
// one pass (it touches only rows/cols within +-2 of its own edge and reads its decision pixels
// before writing), so a pass is a flat parallel loop; vertical edges complete before horizontal.
#pragma once
#include "tk_common.h"

namespace tk {

struct DbParams {
  int width, height, bitdepth;
  int beta, tc_y, tc_c;
  const DbCell* cells;
  int cs;
};

TK_DEV int cell_mvbig(const DbCell& c) {
  return iabs(c.mv0.y) >= 4 || iabs(c.mv0.x) >= 4 || iabs(c.mv1.y) >= 4 || iabs(c.mv1.x) >= 4;
}

// One luma edge segment. dir 0: vertical edge at x = 8*bj (bj >= 1), rows 8*bi..+7;
// dir 1: horizontal edge at y = 8*bi (bi >= 1), cols 8*bj..+7.
template <typename PIX>
TK_DEV void deblock_y_segment(PIX* rec, int stride, const DbParams& P, int bi, int bj, int dir) {
  const int i = bi * 8, j = bj * 8;
  // step along the edge (a) and across it (x)
  const int sa = dir == 0 ? stride : 1;
  const int sx = dir == 0 ? 1 : stride;
  PIX* e = rec + i * stride + j;  // q0 of line 0
  auto dline = [&](int l) -> int {
    const PIX* p = e + l * sa;
    return iabs((int)p[-2 * sx] - (int)p[-sx]) + iabs((int)p[sx] - (int)p[0]);
  };
  const int d15 = dline(1) + dline(5);
  const int d26 = dline(2) + dline(6);
  for (int m = 0; m < 8; m += 4) {
    const int qi = dir == 0 ? ((i + m) / 4) * P.cs + j / 4 : (i / 4) * P.cs + (j + m) / 4;
    const int pi = dir == 0 ? qi - 1 : qi - P.cs;
    const DbCell& q = P.cells[qi];
    const DbCell& p = P.cells[pi];
    int q_size = q.size;
    const int tbs = q.tbpb & 1, pb = q.tbpb >> 1;
    if ((tbs || pb == (dir == 0 ? P_VER : P_HOR) || pb == P_QUAD) && q_size > kMinBlk) q_size /= 2;
    const int mv = cell_mvbig(p) || cell_mvbig(q);
    const int cbp = (p.cbp & 1) || (q.cbp & 1);
    const int mode = p.mode == M_INTRA || q.mode == M_INTRA;
    const int interior = ((dir == 0 ? j : i) % q_size) > 0;
    if (interior || !(mv || cbp || mode)) continue;
    for (int k = m; k < m + 4; k++) {
      const int d = (k & 1) ? d26 : d15;
      if (d >= P.beta) continue;
      PIX* l = e + k * sa;
      const int p1 = l[-2 * sx], p0 = l[-sx], q0 = l[0], q1 = l[sx];
      int delta = (18 * (q0 - p0) - 6 * (q1 - p1) + 16) >> 5;
      delta = clampi(delta, -P.tc_y, P.tc_y);
      l[-2 * sx] = (PIX)sat_pix(p1 + delta / 2, P.bitdepth);
      l[-sx] = (PIX)sat_pix(p0 + delta, P.bitdepth);
      l[0] = (PIX)sat_pix(q0 - delta, P.bitdepth);
      l[sx] = (PIX)sat_pix(q1 - delta / 2, P.bitdepth);
    }
  }
}

// One chroma edge segment (4 chroma samples) of plane `rec` (4:2:0), same (bi,bj,dir) grid as luma.
template <typename PIX>
TK_DEV void deblock_c_segment(PIX* rec, int stride, const DbParams& P, int bi, int bj, int dir) {
  const int i = bi * 8, j = bj * 8;
  const int qi = (i / 4) * P.cs + j / 4;
  const int pi = dir == 0 ? qi - 1 : qi - P.cs;
  const DbCell& q = P.cells[qi];
  const DbCell& p = P.cells[pi];
  const int mode = p.mode == M_INTRA || q.mode == M_INTRA;
  const int interior = ((dir == 0 ? j : i) % (int)q.size) > 0;
  if (interior || !mode) return;
  const int sa = dir == 0 ? stride : 1;
  const int sx = dir == 0 ? 1 : stride;
  PIX* e = rec + (i >> 1) * stride + (j >> 1);
  for (int k = 0; k < 4; k++) {
    PIX* l = e + k * sa;
    const int p1 = l[-2 * sx], p0 = l[-sx], q0 = l[0], q1 = l[sx];
    int delta = (4 * (q0 - p0) + (p1 - q1) + 4) >> 3;
    delta = clampi(delta, -P.tc_c, P.tc_c);
    l[-sx] = (PIX)sat_pix(p0 + delta, P.bitdepth);
    l[0] = (PIX)sat_pix(q0 - delta, P.bitdepth);
  }
}

// pass: 0 = Y vertical, 1 = Y horizontal, 2 = UV vertical, 3 = UV horizontal.
template <typename PIX>
TK_DEV void deblock_pass(const Plane3<PIX>& rec, const DbParams& P, int pass, int gid, int gsize) {
  const int nbi = P.height / 8, nbj = P.width / 8;
  const int dir = pass & 1;
  const int n = nbi * nbj;
  for (int it = gid; it < n; it += gsize) {
    const int bi = it / nbj, bj = it - bi * nbj;
    if (dir == 0 && bj == 0) continue;
    if (dir == 1 && bi == 0) continue;
    if (pass < 2) deblock_y_segment(rec.y, rec.sy, P, bi, bj, dir);
    else {
      deblock_c_segment(rec.u, rec.sc, P, bi, bj, dir);
      deblock_c_segment(rec.v, rec.sc, P, bi, bj, dir);
    }
  }
}

// create_reference_frame: copy + replicate-pad (pad luma 160, chroma 80).
// One work item per padded row (Y rows, then U, then V); `lane`/`nlanes` stride along the row.
template <typename PIX>
TK_DEV void make_ref_rows(const Plane3<PIX>& rec, const Plane3<PIX>& ref, int width, int height, int gid, int gsize,
                          int lane, int nlanes) {
  const int py = kPadY, pc = kPadY / 2;
  const int hy = height + 2 * py, hc = height / 2 + 2 * pc;
  const int total = hy + 2 * hc;
  for (int it = gid; it < total; it += gsize) {
    const PIX* src;
    PIX* dst;
    int w, h, pad, ss, ds, row;
    if (it < hy) { row = it - py; src = rec.y; dst = ref.y; w = width; h = height; pad = py; ss = rec.sy; ds = ref.sy; }
    else if (it < hy + hc) { row = it - hy - pc; src = rec.u; dst = ref.u; w = width / 2; h = height / 2; pad = pc; ss = rec.sc; ds = ref.sc; }
    else { row = it - hy - hc - pc; src = rec.v; dst = ref.v; w = width / 2; h = height / 2; pad = pc; ss = rec.sc; ds = ref.sc; }
    const int sr = clampi(row, 0, h - 1);
    const PIX* s = src + sr * ss;
    PIX* d = dst + row * ds;
    for (int x = -pad + lane; x < w + pad; x += nlanes) d[x] = s[clampi(x, 0, w - 1)];
  }
}

}  // namespace tk
