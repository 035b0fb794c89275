// ubench_icache.cpp - how fast does a gfx950 wave stream through straight-line code that does not fit the
// instruction cache?  (Measurement tool for DESIGN.md: k_superblocks is ~0.3 MB of mostly loop-free code.)
//
// Kernel body: KB kilobytes of 8-byte VALU instructions (v_add_u32 with a 32-bit literal), repeated so that every
// wave issues the same number of instructions whatever KB is.  W waves per CU (one 64-lane block each); with
// `stagger` every wave first idles a different time so that the waves of a CU sit at different places of the body
// (the situation inside a persistent kernel whose waves run unrelated phases).  Reports shader cycles per
// instruction and wave (s_memtime) - 4 is the VALU issue floor of one wave64 on a SIMD-32.
//
//   hipcc --offload-arch=gfx950 -O2 -o ubench_icache tools/ubench_icache.cpp && ./ubench_icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define STR2(x) #x
#define STR(x) STR2(x)

template <int KB> __global__ void __launch_bounds__(64) k_body(unsigned* out, long long* cyc, int reps, int stagger_cycles) {
  unsigned x = threadIdx.x;
  // desynchronise: block b waits (b / 256) * stagger cycles (blocks are dealt round-robin over the CUs, so the
  // W blocks of one CU get W different delays)
  if (stagger_cycles) {
    const long long until = (long long)__builtin_readcyclecounter() + (long long)(blockIdx.x / 256) * stagger_cycles;
    while ((long long)__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(1);
  }
  const long long t0 = (long long)__builtin_readcyclecounter();
  // the loop lives inside the asm statement: the compiler cannot see the size of a .rept body, and bodies beyond
  // 128 KB need a long jump back (s_setpc) instead of a 16-bit branch offset
  asm volatile(
      "s_mov_b32 s20, %2\n"
      "s_getpc_b64 s[22:23]\n"
      ".rept %1\n v_add_u32 %0, 0x12345, %0\n .endr\n"
      "s_sub_u32 s20, s20, 1\n"
      "s_cmp_lg_u32 s20, 0\n"
      "s_cbranch_scc0 2f\n"
      "s_setpc_b64 s[22:23]\n"
      "2:\n"
      : "+v"(x) : "n"(KB * 1024 / 8), "s"(reps) : "s20", "s22", "s23", "scc");
  const long long t1 = (long long)__builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + threadIdx.x] = x;
}

template <int KB> static void run(int W, int stagger, unsigned* d_out, long long* d_cyc) {
  const long long total_instr = 1 << 19;  // per wave
  const int per_rep = KB * 1024 / 8;
  const int reps = (int)(total_instr / per_rep);
  const int blocks = 256 * W;
  // stagger so that the W waves of a CU are spread evenly over one pass of the body (at 4 cycles per instruction)
  const int stag = stagger ? (per_rep * 4) / W + 37 : 0;
  hipLaunchKernelGGL(k_body<KB>, dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, reps, stag);  // warm-up
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k_body<KB>, dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, reps, stag);
  hipDeviceSynchronize();
  std::vector<long long> c(blocks);
  hipMemcpy(c.data(), d_cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double sum = 0, mx = 0;
  for (long long v : c) { sum += (double)v; if ((double)v > mx) mx = (double)v; }
  const double n = (double)reps * per_rep;
  printf("body %4d KB  waves/CU %2d  stagger %d  cycles/instr/wave: mean %7.2f  max %7.2f\n", KB, W, stagger, sum / blocks / n, mx / n);
  fflush(stdout);
}

int main() {
  unsigned* d_out;
  long long* d_cyc;
  hipMalloc(&d_out, 256 * 16 * 64 * sizeof(unsigned));
  hipMalloc(&d_cyc, 256 * 16 * sizeof(long long));
  const int Ws[] = {1, 4, 12};
  for (int stagger = 0; stagger <= 1; stagger++)
    for (int W : Ws) {
      run<8>(W, stagger, d_out, d_cyc);
      run<32>(W, stagger, d_out, d_cyc);
      run<48>(W, stagger, d_out, d_cyc);
      run<64>(W, stagger, d_out, d_cyc);
      run<96>(W, stagger, d_out, d_cyc);
      run<128>(W, stagger, d_out, d_cyc);
      run<256>(W, stagger, d_out, d_cyc);
      run<512>(W, stagger, d_out, d_cyc);
    }
  return 0;
}
