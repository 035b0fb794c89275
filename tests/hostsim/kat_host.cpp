// kat_host.cpp - TEST INFRASTRUCTURE ONLY.  The 1-lane host build of the device functions behind the round-6 known-answer entry points (intra edges +
// prediction, CLPF statistics / filter, CDEF direction / filter), so that tests/golden/kat5.npz pins them on the CPU too (tests/test_kat_host.py);
// the device itself is pinned by tests/test_gpu_kat.py through libthor_hip.so.
#include <cmath>
#include <vector>
using std::sqrt; using std::floor;
#include "../../thor_amd/csrc/tk_pred.h"
#include "../../thor_amd/csrc/tk_clpf.h"
#include "../../thor_amd/csrc/tk_tables.h"
namespace tk { Tables g_tab; }
using namespace tk;
template <typename PIX>
static void intra(const PIX* plane, int stride, int bitdepth, int size, int tb_split, int n, const int* par, const PIX* rblocks, PIX* out) {
  const Team t = mk_team(0, 1);
  for (int it = 0; it < n; it++) {
    static IntraEdge<PIX> edge;
    const int* q = par + 7 * it;
    const int cbs = tb_split ? 2 * size : size;
    const PIX* rblock = tb_split ? rblocks + (size_t)it * cbs * cbs + q[5] * cbs + q[6] : nullptr;
    make_edges<SP_GLOBAL>(t, &edge, plane + (size_t)q[0] * stride + q[1], stride, rblock, cbs, q[5], q[6], q[0], q[1], size, q[2], q[3], tb_split, bitdepth);
    pred_intra<SP_GLOBAL>(t, &edge, q[0] + q[5], q[1] + q[6], size, out + (size_t)it * size * size, size, q[4], bitdepth);
  }
}
extern "C" void h_intra(const void* plane, int stride, int bitdepth, int size, int tb, int n, const int* par, const void* rb, void* out) {
  if (bitdepth == 8) intra<uint8_t>((const uint8_t*)plane, stride, 8, size, tb, n, par, (const uint8_t*)rb, (uint8_t*)out);
  else intra<uint16_t>((const uint16_t*)plane, stride, bitdepth, size, tb, n, par, (const uint16_t*)rb, (uint16_t*)out);
}
template <typename PIX> static void clpf(const PIX* rec, const PIX* org, int w, int h, int bd, int qp, const DbCell* cells, const int* strength, int fb_log2, const uint8_t* fb_on, uint32_t* stats, PIX* out) {
  ClpfJob<PIX> J;
  memset(&J, 0, sizeof(J));
  std::vector<PIX> r(rec, rec + w * h * 3 / 2), s(rec, rec + w * h * 3 / 2), o(org, org + w * h * 3 / 2);
  auto pl = [&](std::vector<PIX>& v) { Plane3<PIX> p; p.y = v.data(); p.u = v.data() + w * h; p.v = p.u + w * h / 4; p.sy = w; p.sc = w / 2; return p; };
  J.rec = pl(r); J.src = pl(s); J.org = pl(o); J.width = w; J.height = h; J.bitdepth = bd; J.qp = qp; J.cells = cells; J.cs = w / 4; J.stats = stats;
  for (int k = 0; k < 3; k++) J.strength[k] = strength[k];
  J.fb_log2 = fb_log2; J.fb_on = fb_on;
  clpf_pass_stats(J, 0, 1);
  clpf_pass_apply(J, 0, 1);
  memcpy(out, r.data(), sizeof(PIX) * r.size());
}
extern "C" void h_clpf(const void* rec, const void* org, int w, int h, int bd, int qp, const void* cells, const int* strength, int fb_log2, const uint8_t* fb_on, uint32_t* stats, void* out) {
  if (bd == 8) clpf<uint8_t>((const uint8_t*)rec, (const uint8_t*)org, w, h, 8, qp, (const DbCell*)cells, strength, fb_log2, fb_on, stats, (uint8_t*)out);
  else clpf<uint16_t>((const uint16_t*)rec, (const uint16_t*)org, w, h, bd, qp, (const DbCell*)cells, strength, fb_log2, fb_on, stats, (uint16_t*)out);
}

extern "C" void h_cdef_dir(const void* blocks, int bd, int n, int* dir, int* var) {
  for (int i = 0; i < n; i++) {
    int v = 0;
    dir[i] = bd == 8 ? cdef_find_dir((const uint8_t*)blocks + (size_t)i * 64, 8, &v, 0) : cdef_find_dir((const uint16_t*)blocks + (size_t)i * 64, 8, &v, bd - 8);
    var[i] = v;
  }
}
extern "C" void h_cdef_filter(const void* plane, int w, int h, int stride, int bd, int bsize, int n, const int* par, void* out) {
  for (int it = 0; it < n; it++) {
    const int* q = par + 7 * it;
    for (int k = 0; k < bsize * bsize; k++) {
      const int x = q[0] + k % bsize, y = q[1] + k / bsize;
      if (bd == 8) ((uint8_t*)out)[(size_t)it * bsize * bsize + k] = (uint8_t)cdef_filter_px((const uint8_t*)plane, stride, x, y, w, h, q[2], q[3], q[4], q[5], q[6], 0);
      else ((uint16_t*)out)[(size_t)it * bsize * bsize + k] = (uint16_t)cdef_filter_px((const uint16_t*)plane, stride, x, y, w, h, q[2], q[3], q[4], q[5], q[6], bd - 8);
    }
  }
}
