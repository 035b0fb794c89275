// ubench_l1gather.cpp - throughput of the vector L1 (TCP) for the gather patterns of the motion search, all hits.
// Every wave re-reads its own small region of a padded 8-bit plane (pitch 4160) so that the data stays in the CU's 32 KiB L1;
// 12 waves per CU (the residency of k_superblocks).  Patterns, one global_load per lane and iteration:
//   0  coalesced     16 bytes per lane, 1 KiB contiguous per wave-instruction                    (reference point)
//   1  seg16 rows    16 bytes per lane, 4 lanes per 64-byte block row, 16 rows, byte offset 5      (large-PU candidate rows)
//   2  seg8 rows     8 bytes per lane, one lane per row of an 8x8 block, 8 blocks side by side     (small-PU candidate sets)
//   3  dword gather  4 bytes per lane, one lane per row (round 2's pattern)
// Each pattern at 1, 4 and 12 waves per CU with 1 or 4 independent loads in flight per wave: the first line is the latency of one
// instruction, the last the throughput.  Prints cycles per wave-instruction and lane-bytes per clock and CU.
//   hipcc --offload-arch=gfx950 -O2 -o ubench_l1gather tools/ubench_l1gather.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u4u;
typedef uint32_t __attribute__((ext_vector_type(2), aligned(1))) u2u;
typedef uint32_t __attribute__((aligned(1))) u1u;

template <int PAT, int U> __global__ void __launch_bounds__(64) k(const uint8_t* plane, int iters, long long* cyc, unsigned* sink) {
  const int lane = threadIdx.x, pitch = 4160;
  const uint8_t* base = plane + (size_t)(blockIdx.x % 3072) * 24 * pitch + 160;   // the wave's own rows
  unsigned acc = 0;
  const long long t0 = (long long)__builtin_readcyclecounter();
  for (int it = 0; it < iters; it += U) {
    unsigned r[U];
#pragma unroll
    for (int u = 0; u < U; u++) {   // U independent loads in flight per wave
      const int dy = (it + u) & 3, dx = ((it + u) >> 2) & 7;   // candidate displacement: the wave touches <= 20 rows x one 128-byte line (12 waves fit the L1)
      if (PAT == 0) { const u4u v = *(const u4u*)(base + (((it >> 2) + u) & 1) * pitch + lane * 16); r[u] = v.x ^ v.w; }   // 2 rows x 1 KiB per wave
      if (PAT == 1) { const u4u v = *(const u4u*)(base + ((lane >> 2) + dy) * pitch + (lane & 3) * 16 + dx + 5); r[u] = v.x ^ v.w; }
      if (PAT == 2) { const u2u v = *(const u2u*)(base + ((lane & 7) + dy) * pitch + (lane >> 3) * 9 + dx + 3); r[u] = v.x ^ v.y; }
      if (PAT == 3) { const u1u v = *(const u1u*)(base + ((lane & 15) + dy) * pitch + (lane >> 4) * 13 + dx + 1); r[u] = v; }
    }
#pragma unroll
    for (int u = 0; u < U; u++) acc += r[u];
  }
  const long long t1 = (long long)__builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int PAT, int U> static void run1(const char* name, int bytes_per_lane, int W, const uint8_t* d, long long* d_cyc, unsigned* d_sink) {
  const int blocks = 256 * W, iters = 20000;
  hipLaunchKernelGGL((k<PAT, U>), dim3(blocks), dim3(64), 0, 0, d, iters, d_cyc, d_sink);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL((k<PAT, U>), dim3(blocks), dim3(64), 0, 0, d, iters, d_cyc, d_sink);
  (void)hipDeviceSynchronize();
  std::vector<long long> c(blocks);
  (void)hipMemcpy(c.data(), d_cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double sum = 0;
  for (long long v : c) sum += (double)v;
  const double per = sum / blocks / iters;   // cycles per wave-instruction as seen by one wave (W waves share the L1)
  printf("%-14s %2d B/lane  %2d waves/CU x %d in flight: %7.1f cycles per instruction and wave -> one per %6.1f cycles per CU, %6.1f lane-bytes/clk/CU\n", name,
         bytes_per_lane, W, U, per, per / W, 64.0 * bytes_per_lane * W / per);
}
template <int PAT> static void run(const char* name, int bytes_per_lane, const uint8_t* d, long long* d_cyc, unsigned* d_sink) {
  run1<PAT, 1>(name, bytes_per_lane, 1, d, d_cyc, d_sink);    // latency of one instruction
  run1<PAT, 4>(name, bytes_per_lane, 1, d, d_cyc, d_sink);
  run1<PAT, 1>(name, bytes_per_lane, 4, d, d_cyc, d_sink);
  run1<PAT, 4>(name, bytes_per_lane, 4, d, d_cyc, d_sink);
  run1<PAT, 1>(name, bytes_per_lane, 12, d, d_cyc, d_sink);   // the residency of k_superblocks
  run1<PAT, 4>(name, bytes_per_lane, 12, d, d_cyc, d_sink);   // throughput
}

int main() {
  uint8_t* d;
  long long* d_cyc;
  unsigned* d_sink;
  const size_t bytes = (size_t)3072 * 24 * 4160 + (1 << 20);
  (void)hipMalloc(&d, bytes);
  (void)hipMemset(d, 7, bytes);
  (void)hipMalloc(&d_cyc, 3072 * sizeof(long long));
  (void)hipMalloc(&d_sink, 4);
  run<0>("coalesced", 16, d, d_cyc, d_sink);
  run<1>("seg16 rows", 16, d, d_cyc, d_sink);
  run<2>("seg8 rows", 8, d, d_cyc, d_sink);
  run<3>("dword gather", 4, d, d_cyc, d_sink);
  return 0;
}
