// hostsim.cpp - TEST INFRASTRUCTURE ONLY.
// Compiles the engine headers with -DTHOR_HOSTSIM (1-lane teams, plain loops) into a CPU
// executable so that the bit-exactness of the algorithm can be checked against the reference
// encoder (oracle/_ref/Thorenc) in a container without a GPU.  The product library
// (thor_amd/csrc/thor_hip.cpp) never links or calls this; it has no CPU path.
//
// With -DTHOR_HOSTSIM_LANES=N (N a power of two <= 64, e.g. 8) every superblock is processed by a team of N lanes, each lane
// an OS thread executing the same SPMD code as a GPU lane; barriers, ballots, shuffles and reductions go through one
// "every lane publishes a value, all lanes read" exchange.  This exercises the lane-parallel logic of the engine (work
// distribution, reductions, the ballot automata, uniformity assumptions - tk_uniform() aborts if lanes disagree) without a
// GPU; a lane that skips a collective deadlocks the run, which is the bug being looked for.
#include "../../thor_amd/csrc/tk_block.h"
#include "../../thor_amd/csrc/tk_filters.h"
#include <math.h>
#include "../../thor_amd/csrc/tk_cli.h"

namespace tk {
Tables g_tab;
namespace backend {
void* dev_alloc(size_t n) { return calloc(1, n); }
void dev_free(void* p) { free(p); }
void h2d(void* d, const void* h, size_t n) { memcpy(d, h, n); }
void d2h(void* h, const void* d, size_t n) { memcpy(h, d, n); }
void dev_memset(void* d, int v, size_t n) { memset(d, v, n); }
void dev_sync() {}
void release_superblocks(const void*) {}
size_t team_ws_bytes(int pix_bytes) {  // one workgroup: shared state + kWaves x (BigWs + SmallWs)
  return sizeof(WgShared) + 64 + (size_t)kWaves * (pix_bytes == 1 ? sizeof(BigWs<uint8_t>) + sizeof(SmallWs<uint8_t>) + 128 : sizeof(BigWs<uint16_t>) + sizeof(SmallWs<uint16_t>) + 128);
}
// carve the workgroup's scratch arena: [WgShared][wave 0: BigWs SmallWs][wave 1: ...]
template <typename PIX> static TeamWs<PIX> wave_ws(uint8_t* base, int wave) {
  auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
  WgShared* sh = (WgShared*)base;
  uint8_t* p = base + al(sizeof(WgShared)) + (size_t)wave * (al(sizeof(BigWs<PIX>)) + al(sizeof(SmallWs<PIX>)));
  return make_ws((SmallWs<PIX>*)(p + al(sizeof(BigWs<PIX>))), sh, (BigWs<PIX>*)p);
}

#ifdef THOR_HOSTSIM_LANES
}  // namespace backend
}  // namespace tk
#include <atomic>
#include <thread>
#include <vector>
namespace tk {
namespace hostlanes {
struct Shared {
  int n = 1;
  std::atomic<int> count{0};
  std::atomic<int> sense{0};
  unsigned long long slots[64];
};
static thread_local Shared* tl_sh = nullptr;
static thread_local int tl_rank = 0;
static thread_local int tl_sense = 0;
int lanes() { return tl_sh ? tl_sh->n : 1; }
int rank() { return tl_rank; }
void barrier() {
  Shared* sh = tl_sh;
  if (!sh || sh->n == 1) return;
  const int my = tl_sense ^= 1;
  if (sh->count.fetch_add(1, std::memory_order_acq_rel) == sh->n - 1) {
    sh->count.store(0, std::memory_order_relaxed);
    sh->sense.store(my, std::memory_order_release);
  } else {
    int spins = 0;
    while (sh->sense.load(std::memory_order_acquire) != my)
      if (++spins > 2000) { std::this_thread::yield(); spins = 0; }
  }
}
static unsigned long long tl_single[1];
const unsigned long long* exchange_begin(unsigned long long v) {
  Shared* sh = tl_sh;
  if (!sh || sh->n == 1) { tl_single[0] = v; return tl_single; }
  sh->slots[tl_rank] = v;
  barrier();
  return sh->slots;
}
void exchange_end() { barrier(); }
}  // namespace hostlanes
namespace backend {
template <typename PIX> void run_superblocks(const FrameJob<PIX>* jobs, const FrameJob<PIX>*, int S, const SbRange* ranges) {
  const int N = THOR_HOSTSIM_LANES;
  for (int s = 0; s < S; s++) {
    if (ranges && ranges[s].lo >= ranges[s].hi) continue;   // nothing of this stream in the launch (its job may not be set up yet)
    const FrameJob<PIX>& J = jobs[s];
    for (int k = 0; k < J.sb_rows; k++)
      for (int l = 0; l < J.sb_cols; l++) {
        if (ranges && !(l + 2 * k >= ranges[s].lo && l + 2 * k < ranges[s].hi)) continue;   // raster order inside a range respects the dependencies
        const int sbi = k * J.sb_cols + l;
        hostlanes::Shared sh;
        sh.n = N;
        std::vector<std::thread> th;
        for (int r = 0; r < N; r++)
          th.emplace_back([&, r]() {
            hostlanes::tl_sh = &sh; hostlanes::tl_rank = r; hostlanes::tl_sense = 0;
            TeamWs<PIX> wsv = wave_ws<PIX>(J.scratch, 0);
            Team t{r, N, wsv.sh->tabs.izz};
            xform_tables_fill(&wsv.sh->tabs, r, N);
            t.sync();
            BitSink out;
            out.buf = J.sb_bits + (size_t)sbi * J.sb_words; out.pos = 0; out.cap = J.sb_words * 32; out.emit = 1; out.ovf = 0;
            process_sb(Wg{0, 1}, t, J, &wsv, k * kMaxSb, l * kMaxSb, out);
            if (r == 0) { J.sb_nbits[sbi] = out.pos; J.sb_status[sbi] = out.ovf; }
          });
        for (auto& x : th) x.join();
      }
  }
}
#elif defined(THOR_HOSTSIM_WAVES)
}  // namespace backend
}  // namespace tk
#include <atomic>
#include <thread>
#include <vector>
// Multi-wave workgroups: every wavefront of a workgroup is an OS thread with a 1-lane team; the workgroup barrier is a
// sense-reversing thread barrier and the shared state is accessed with the same atomics as on the device.  This runs
// the fork/join protocol, the work queue and the key-based pruning of mode_decision_par under real concurrency.
namespace tk {
namespace hostwaves {
struct Bar { int n; std::atomic<int> count{0}; std::atomic<int> sense{0}; };
static thread_local Bar* tl_bar = nullptr;
static thread_local int tl_sense = 0;
void barrier() {
  Bar* b = tl_bar;
  if (!b || b->n == 1) return;
  const int my = tl_sense ^= 1;
  if (b->count.fetch_add(1, std::memory_order_acq_rel) == b->n - 1) {
    b->count.store(0, std::memory_order_relaxed);
    b->sense.store(my, std::memory_order_release);
  } else {
    int spins = 0;
    while (b->sense.load(std::memory_order_acquire) != my)
      if (++spins > 200) { std::this_thread::yield(); spins = 0; }
  }
}
}  // namespace hostwaves
namespace backend {
template <typename PIX> void run_superblocks(const FrameJob<PIX>* jobs, const FrameJob<PIX>*, int S, const SbRange* ranges) {
  const int NW = kWaves;
  for (int s = 0; s < S; s++) {
    if (ranges && ranges[s].lo >= ranges[s].hi) continue;
    const FrameJob<PIX>& J = jobs[s];
    hostwaves::Bar bar;
    bar.n = NW;
    xform_tables_fill(&((WgShared*)J.scratch)->tabs, 0, 1);
    std::vector<std::thread> th;
    for (int w = 0; w < NW; w++)
      th.emplace_back([&, w]() {
        hostwaves::tl_bar = &bar; hostwaves::tl_sense = 0;
        const Wg wg{w, NW};
        TeamWs<PIX> wsv = wave_ws<PIX>(J.scratch, w);
        const Team t{0, 1, wsv.sh->tabs.izz};
        for (int k = 0; k < J.sb_rows; k++)
          for (int l = 0; l < J.sb_cols; l++) {
            if (ranges && !(l + 2 * k >= ranges[s].lo && l + 2 * k < ranges[s].hi)) continue;
            const int sbi = k * J.sb_cols + l;
            if (w == 0) {
              BitSink out;
              out.buf = J.sb_bits + (size_t)sbi * J.sb_words; out.pos = 0; out.cap = J.sb_words * 32; out.emit = 1; out.ovf = 0;
              process_sb(wg, t, J, &wsv, k * kMaxSb, l * kMaxSb, out);
              J.sb_nbits[sbi] = out.pos;
              J.sb_status[sbi] = out.ovf;
            } else
              wg_helper_loop(wg, t, J, &wsv);
          }
      });
    for (auto& x : th) x.join();
  }
}
#else
template <typename PIX> void run_superblocks(const FrameJob<PIX>* jobs, const FrameJob<PIX>*, int S, const SbRange* ranges) {
  for (int s = 0; s < S; s++) {
    if (ranges && ranges[s].lo >= ranges[s].hi) continue;
    const FrameJob<PIX>& J = jobs[s];
    TeamWs<PIX> wsv = wave_ws<PIX>(J.scratch, 0);
    TeamWs<PIX>* ws = &wsv;
    xform_tables_fill(&ws->sh->tabs, 0, 1);
    const Team t{0, 1, ws->sh->tabs.izz};
    for (int k = 0; k < J.sb_rows; k++)
      for (int l = 0; l < J.sb_cols; l++) {
        if (ranges && !(l + 2 * k >= ranges[s].lo && l + 2 * k < ranges[s].hi)) continue;
        const int sbi = k * J.sb_cols + l;
        BitSink out;
        out.buf = J.sb_bits + (size_t)sbi * J.sb_words; out.pos = 0; out.cap = J.sb_words * 32; out.emit = 1; out.ovf = 0;
        process_sb(Wg{0, 1}, t, J, ws, k * kMaxSb, l * kMaxSb, out);
        J.sb_nbits[sbi] = out.pos;
        J.sb_status[sbi] = out.ovf;
      }
  }
}
#endif
template <typename PIX> void run_deblock(const FrameJob<PIX>* jobs, const FrameJob<PIX>*, int S) {
  for (int s = 0; s < S; s++) {
    const FrameJob<PIX>& J = jobs[s];
    DbParams P;
    P.width = J.cfg.width; P.height = J.cfg.height; P.bitdepth = J.cfg.bitdepth;
    const int qpc = g_tab.chroma_qp[J.qp];
    P.beta = g_tab.beta[J.qp] << (P.bitdepth - 8);
    P.tc_y = g_tab.tc[J.qp] >> (12 - P.bitdepth);
    P.tc_c = g_tab.tc[qpc] >> (12 - P.bitdepth);
    P.cells = J.cells; P.cs = J.cell_stride;
    for (int pass = 0; pass < 4; pass++) deblock_pass(J.rec, P, pass, 0, 1);
  }
}
template <typename PIX> void run_make_ref(const FrameJob<PIX>* hjobs, const Plane3<PIX>* dst, int S) {
  for (int s = 0; s < S; s++) make_ref_rows(hjobs[s].rec, dst[s], hjobs[s].cfg.width, hjobs[s].cfg.height, 0, 1, 0, 1);
}
void run_gather(const GatherItem* items, int n, uint32_t* dst) {
  for (int it = 0; it < n; it++) {
    const GatherItem& g = items[it];
    const int nw = (g.nbits + 31) >> 5, sh = (int)(g.dst_bit & 31);
    const long long w0 = g.dst_bit >> 5;
    for (int j = 0; j < nw; j++) {
      uint32_t v = g.src[j];
      const int valid = g.nbits - 32 * j;
      if (valid < 32) v &= ~((1u << (32 - valid)) - 1u);
      if (sh == 0) dst[w0 + j] |= v;
      else { dst[w0 + j] |= v >> sh; dst[w0 + j + 1] |= v << (32 - sh); }
    }
  }
}
template <typename PIX> void run_cdef(const CdefJob<PIX>* cj, const CdefJob<PIX>*, int S) {
  Team t{0, 1};
  for (int s = 0; s < S; s++) {
    const CdefJob<PIX>& C = cj[s];
    for (int i = 0; i < C.height; i++) memcpy(C.src.y + (size_t)i * C.src.sy, C.rec.y + (size_t)i * C.rec.sy, C.width * sizeof(PIX));
    for (int i = 0; i < C.height / 2; i++) {
      memcpy(C.src.u + (size_t)i * C.src.sc, C.rec.u + (size_t)i * C.rec.sc, C.width / 2 * sizeof(PIX));
      memcpy(C.src.v + (size_t)i * C.src.sc, C.rec.v + (size_t)i * C.rec.sc, C.width / 2 * sizeof(PIX));
    }
    cdef_pass_flags(C, 0, 1);
    cdef_pass_dir(C, 0, 1);
    if (C.cdef_bits) {   // the wavefront form the device runs (lanes as a loop); the plain form is compared with it in tests/hostsim/unit_cdef.cpp
      static CdefWaveWs<PIX> ws;
      for (int b = 0; b < (C.width / 8) * (C.height / 8); b++) cdef_mse_block_wave(t, C, b, &ws);
    }
    cdef_pass_select(t, C);
    cdef_pass_apply(C, 0, 1);
  }
}
template <typename PIX> void run_clpf_stats(const ClpfJob<PIX>* lj, const ClpfJob<PIX>*, int S) {
  for (int s = 0; s < S; s++) clpf_pass_stats(lj[s], 0, 1);
}
template <typename PIX> void run_clpf_apply(const ClpfJob<PIX>* lj, const ClpfJob<PIX>*, int S) {
  for (int s = 0; s < S; s++) {
    const ClpfJob<PIX>& C = lj[s];
    for (int i = 0; i < C.height; i++) memcpy(C.src.y + (size_t)i * C.src.sy, C.rec.y + (size_t)i * C.rec.sy, C.width * sizeof(PIX));
    for (int i = 0; i < C.height / 2; i++) {
      memcpy(C.src.u + (size_t)i * C.src.sc, C.rec.u + (size_t)i * C.rec.sc, C.width / 2 * sizeof(PIX));
      memcpy(C.src.v + (size_t)i * C.src.sc, C.rec.v + (size_t)i * C.rec.sc, C.width / 2 * sizeof(PIX));
    }
    clpf_pass_apply(C, 0, 1);
  }
}
// temporal interpolation: the device phases as plain loops (1-lane teams, block rows in raster order)
template <typename PIX> void run_interp(const idev::Job<PIX>* jobs, const idev::Job<PIX>*, int n) {
  using namespace idev;
  const Team t{0, 1};
  for (int s = 0; s < n; s++) {
    const Job<PIX>& J = jobs[s];
    for (int l = 0; l < J.levels; l++) {
      const Level<PIX>& L = J.lv[l];
      const size_t cnt = (size_t)L.bw * L.bh + L.bw + 2;
      memset(L.mv[0], 0, cnt * sizeof(imv)); memset(L.mv[1], 0, cnt * sizeof(imv));
      memset(L.prog, 0, (size_t)(L.bh / kStep + 1) * sizeof(int));
    }
    for (int l = 1; l < J.levels; l++)
      for (int r = 0; r < 2; r++) {
        const PIX* in = l == 1 ? J.ref[r].y : J.dpic[r][l - 1];
        const int in_s = l == 1 ? J.ref[r].sy : J.dstride[l - 1];
        const int ow = J.width >> l, oh = J.height >> l;
        for (int i = -32; i < oh + 32; i++)
          for (int j = -32; j < ow + 32; j++) down2x2_item(in, in_s, J.dpic[r][l], J.dstride[l], ow, oh, i, j);
      }
    for (int lvl = J.levels - 1; lvl >= 0; --lvl) {
      const Level<PIX>& L = J.lv[lvl];
      for (int i = 0; i < L.bh; i += kStep)
        for (int j = 0; j < L.bw; j += kStep) estimate_block(t, L, i, j);
      for (int i = 0; i < L.bh; i++)
        for (int j = 0; j < L.bw; j++) merge_block(t, L, i, j);
      if (lvl > 0) {
        const Level<PIX>& O = J.lv[lvl - 1];
        for (int i = 0; i < O.bh; i++)
          for (int j = 0; j < O.bw; j++) upscale_item(L.nmv[1], L.bw, O.gmv1, O.bw, i, j);
      } else {
        for (int yp = 0; yp < L.bh; yp++)
          for (int xp = 0; xp < L.bw; xp++) mot_comp_unit(t, J, yp, xp);
        const int rows = J.height + 2 * kPadY + 2 * (J.height / 2 + kPadY);
        for (int it = 0; it < rows; it++) pad_item(J, it, 0, 1);
      }
    }
  }
}
template void run_interp<uint8_t>(const idev::Job<uint8_t>*, const idev::Job<uint8_t>*, int);
template void run_interp<uint16_t>(const idev::Job<uint16_t>*, const idev::Job<uint16_t>*, int);
template void run_clpf_stats<uint8_t>(const ClpfJob<uint8_t>*, const ClpfJob<uint8_t>*, int);
template void run_clpf_stats<uint16_t>(const ClpfJob<uint16_t>*, const ClpfJob<uint16_t>*, int);
template void run_clpf_apply<uint8_t>(const ClpfJob<uint8_t>*, const ClpfJob<uint8_t>*, int);
template void run_clpf_apply<uint16_t>(const ClpfJob<uint16_t>*, const ClpfJob<uint16_t>*, int);
template void run_cdef<uint8_t>(const CdefJob<uint8_t>*, const CdefJob<uint8_t>*, int);
template void run_cdef<uint16_t>(const CdefJob<uint16_t>*, const CdefJob<uint16_t>*, int);
template void run_superblocks<uint8_t>(const FrameJob<uint8_t>*, const FrameJob<uint8_t>*, int, const SbRange*);
template void run_superblocks<uint16_t>(const FrameJob<uint16_t>*, const FrameJob<uint16_t>*, int, const SbRange*);
template void run_deblock<uint8_t>(const FrameJob<uint8_t>*, const FrameJob<uint8_t>*, int);
template void run_deblock<uint16_t>(const FrameJob<uint16_t>*, const FrameJob<uint16_t>*, int);
template void run_make_ref<uint8_t>(const FrameJob<uint8_t>*, const Plane3<uint8_t>*, int);
template void run_make_ref<uint16_t>(const FrameJob<uint16_t>*, const Plane3<uint16_t>*, int);
}  // namespace backend
}  // namespace tk

namespace tk { long long g_prune_stat[8]; }
int main(int argc, char** argv) {
  struct R { ~R() { if (getenv("THOR_PRUNE_STAT")) fprintf(stderr, "prune: intra %lld/%lld inter %lld/%lld quadrant checks %lld early %lld at-4th %lld at-entry %lld\n", tk::g_prune_stat[1], tk::g_prune_stat[0], tk::g_prune_stat[3], tk::g_prune_stat[2], tk::g_prune_stat[4], tk::g_prune_stat[5], tk::g_prune_stat[6], tk::g_prune_stat[7]); } } r_;
  tk::init_tables(&tk::g_tab);
  tk::CliArgs a = tk::cli_parse(argc, argv);
  if (a.sp.bitdepth != a.sp.input_bitdepth) { fprintf(stderr, "bitdepth != input_bitdepth is not supported\n"); return 2; }
  return a.sp.bitdepth > 8 ? tk::cli_run<uint16_t>(a) : tk::cli_run<uint8_t>(a);
}
