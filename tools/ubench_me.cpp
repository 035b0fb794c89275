// ubench_me.cpp - motion_estimate (thor_amd/csrc/tk_me.h, the product's function, 8-bit, LDS-resident original block) in isolation: cycles per
// call for PUs of 4x4 .. 32x32 with 1, 2 and 3 workgroups of four waves per CU (the residency of k_superblocks), every wave searching its own
// blocks of a synthetic textured plane against a displaced copy.  What a search costs when nothing else runs on the CU and how much of the
// in-kernel figure (60 k cycles per call for 8x8 coding blocks, profiles/r04_prof_ldb_me_1080p_n14_final_call14.log) is the search itself.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/ubench_me tools/ubench_me.cpp
#include "../thor_amd/csrc/tk_me.h"
#include <cstdio>
#include <vector>
using namespace tk;

enum { W = 1024, H = 512, PITCH = W + 2 * kPadY };

template <int PW, int PH, int CB, int SPACE = SP_LDS> __global__ __launch_bounds__(256, 3) void k_me(const uint8_t* cur, const uint8_t* ref, int iters, long long* cyc, int* sink, long long* prof_out) {
  extern __shared__ uint8_t pad_[];   // dynamic LDS: pads the workgroup to the product's footprint so that at most three fit a CU
  __shared__ MeWs ws[4];
  __shared__ MeLists lists[4];
  __shared__ uint32_t win[4][1200];
  __shared__ uint8_t org[4][32 * 32];
  __shared__ long long prof[4][32];   // -DTHOR_PROF builds: the function's own phase counters
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
  const Team t = mk_team(lane, 64);
  if (lane == 0) {
    for (int q = 0; q < 32; q++) prof[wave][q] = 0;
    ws[wave].lists = &lists[wave]; ws[wave].prof = prof[wave]; ws[wave].win = win[wave]; ws[wave].win_cap = 4500; ws[wave].cwin_valid = 0;
    for (int r = 0; r < kMaxRefs; r++) { lists[wave].mvcand_num[r] = 0; lists[wave].mvcand_mask[r] = 0; }
    pad_[0] = 0;
  }
  __syncthreads();
  const uint8_t* cur0 = cur + (size_t)kPadY * PITCH + kPadY;
  const uint8_t* ref0 = ref + (size_t)kPadY * PITCH + kPadY;
  MeArgs a;
  a.cb_size = CB; a.ostride = PW; a.width = PW; a.height = PH; a.rstride = PITCH; a.sign = 0; a.fwidth = W; a.fheight = H;
  a.enable_bipred = 1; a.bitdepth = 8; a.speed = 0; a.lam = 9.5;
  long long total = 0;
  int acc = 0;
  const int gw = blockIdx.x * 4 + wave;
  for (int it = 0; it < iters; it++) {
    // the wave's next block: walks over the plane, 16-aligned positions away from the frame edge
    const int bx = 64 + 16 * ((gw * 7 + it * 3) % ((W - 160) / 16)), by = 64 + 16 * ((gw * 5 + it) % ((H - 160) / 16));
    if (SPACE == SP_LDS) for (int k = lane; k < PW * PH; k += 64) org[wave][k & 1023] = cur0[(by + k / PW) * PITCH + bx + k % PW];
    if (lane == 0 && (it & 7) == 0) { lists[wave].mvcand_num[0] = 0; lists[wave].mvcand_mask[0] = 0; }   // a new "superblock": empty candidate list
    t.sync();
    a.xpos = bx; a.ypos = by; a.pu_x = bx; a.pu_y = by;
    mv_t mv;
    // predictor: the true motion (20, -12 quarter-pels) in three searches of four, 3 px / 2 px off in the fourth
    const mv_t mvp = (it & 3) ? mk_mv(20, -12) : mk_mv(8, -4);
    const long long t0 = (long long)__builtin_readcyclecounter();
    if (SPACE == SP_LDS) acc += (int)motion_estimate<uint8_t, SP_LDS>(t, &ws[wave], org[wave], ref0 + by * PITCH + bx, a, mvp, mvp, 0, &mv);
    else { a.ostride = PITCH; acc += (int)motion_estimate<uint8_t, SP_GLOBAL>(t, &ws[wave], cur0 + by * PITCH + bx, ref0 + by * PITCH + bx, a, mvp, mvp, 0, &mv); }
    total += (long long)__builtin_readcyclecounter() - t0;
    if (lane == 0) add_mvcand(&ws[wave], 0, mv);
    t.sync();
    acc += mv.x * 3 + mv.y;
  }
  if (lane == 0) cyc[gw] = total;
  if (prof_out && lane < 32) prof_out[(size_t)gw * 32 + lane] = prof[wave][lane];
  if (acc == 0x7fffffff) sink[0] = acc;
}

static long long* d_prof = nullptr;
template <int PW, int PH, int CB, int SPACE = SP_LDS> static void run(const uint8_t* d_cur, const uint8_t* d_ref, long long* d_cyc, int* d_sink) {
  const int iters = PW >= 64 ? 100 : 400;
  for (int per_cu = 1; per_cu <= 3; per_cu++) {
    const int blocks = 256 * per_cu;
    const size_t dyn = 36 * 1024;   // + ~17 KB static: 53 KB per workgroup, three per CU at most
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL((k_me<PW, PH, CB, SPACE>), dim3(blocks), dim3(256), dyn, 0, d_cur, d_ref, iters, d_cyc, d_sink, d_prof);
      if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return; }
    }
    std::vector<long long> c(blocks * 4);
    (void)hipMemcpy(c.data(), d_cyc, c.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (long long v : c) sum += (double)v;
    printf("PU %2dx%-2d (CB %3d): %d workgroup(s) x 4 waves per CU: %8.0f cycles per motion_estimate call\n", PW, PH, CB, per_cu, sum / c.size() / iters);
#ifdef THOR_PROF
    if (per_cu == 1) {
      std::vector<long long> pr((size_t)blocks * 4 * 32);
      (void)hipMemcpy(pr.data(), d_prof, pr.size() * sizeof(long long), hipMemcpyDeviceToHost);
      double sl[32] = {0};
      for (size_t w = 0; w < (size_t)blocks * 4; w++) for (int q = 0; q < 32; q++) sl[q] += (double)pr[w * 32 + q];
      const double n = (double)blocks * 4 * iters;
      printf("    per call: full-pel %.0f (telescope %.0f, candidate list %.0f, hexagon %.0f), sub-pel %.0f (set-up %.0f, sample loop %.0f, reduction + cost %.0f)\n", sl[2] / n,
             sl[13] / n, sl[14] / n, sl[15] / n, sl[3] / n, sl[6] / n, sl[10] / n, sl[7] / n);
    }
#endif
  }
}

int main() {
  std::vector<uint8_t> cur((size_t)PITCH * (H + 2 * kPadY)), ref(cur.size());
  // texture: smooth + fine detail; cur = ref displaced by (5, -3) samples plus a little noise
  unsigned rng = 12345u;
  auto tex = [](int x, int y) -> int { return 128 + (int)(50.0 * sin(x * 0.07) * cos(y * 0.05) + 30.0 * sin(x * 0.31 + y * 0.23) + 14.0 * sin(x * 1.3) * sin(y * 1.1)); };
  for (int y = 0; y < H + 2 * kPadY; y++)
    for (int x = 0; x < PITCH; x++) {
      rng = rng * 1664525u + 1013904223u;
      const int n = (int)((rng >> 24) & 3) - 1;
      ref[(size_t)y * PITCH + x] = (uint8_t)tex(x, y);
      int v = tex(x + 5, y - 3) + n;
      cur[(size_t)y * PITCH + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  uint8_t *d_cur, *d_ref; long long* d_cyc; int* d_sink;
  (void)hipMalloc(&d_cur, cur.size()); (void)hipMalloc(&d_ref, ref.size()); (void)hipMalloc(&d_cyc, 768 * 4 * sizeof(long long)); (void)hipMalloc(&d_sink, 4); (void)hipMalloc(&d_prof, 768 * 4 * 32 * sizeof(long long));
  (void)hipMemcpy(d_cur, cur.data(), cur.size(), hipMemcpyHostToDevice); (void)hipMemcpy(d_ref, ref.data(), ref.size(), hipMemcpyHostToDevice);
  run<4, 4, 8>(d_cur, d_ref, d_cyc, d_sink);
  run<8, 8, 8>(d_cur, d_ref, d_cyc, d_sink);
  run<16, 16, 16>(d_cur, d_ref, d_cyc, d_sink);
  run<32, 32, 32>(d_cur, d_ref, d_cyc, d_sink);
  run<64, 64, 64, SP_GLOBAL>(d_cur, d_ref, d_cyc, d_sink);
  return 0;
}
