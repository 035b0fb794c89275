// ubench_rewrite.cpp - what do FETCH_SIZE / WRITE_SIZE count when a workgroup keeps re-writing and re-reading a SMALL working set of its own
// (the trial buffers and the private segment of k_superblocks)?  768 workgroups of 256 lanes, like the persistent kernel; every workgroup owns
// a region of R bytes and passes over it N times (store, then load what the previous pass stored):
//   k_rewrite<global>  R = 16 KiB per workgroup (12 MiB in total: fits the L2s)      - a write-back L2 keeps all of it: ~12 MiB leave the L2
//   k_rewrite<global>  R = 64 KiB per workgroup (48 MiB: 6 MiB per XCD > its 4 MiB L2, < the 256 MiB Infinity Cache)
//   k_rewrite<global>  R = 512 KiB per workgroup (384 MiB > the Infinity Cache: what the private segments of 768 workgroups add up to)
//   k_private          a 16-dword private array per lane, indexed at run time (scratch memory), N passes
// Compare the counters (KiB) per kernel with the bytes printed here.  Run under `rocprofv3 --pmc WRITE_SIZE --kernel-trace` and `--pmc FETCH_SIZE --kernel-trace`.
//   hipcc --offload-arch=gfx950 -O2 -o ubench_rewrite tools/ubench_rewrite.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(256) k_rewrite(uint4* base, size_t region16, int passes, unsigned* sink) {
  uint4* p = base + (size_t)blockIdx.x * region16;
  unsigned acc = 0;
  for (int n = 0; n < passes; n++) {
    for (size_t i = threadIdx.x; i < region16; i += 256) {
      uint4 v = p[i];                                   // what the previous pass stored
      acc += v.x + v.w;
      p[i] = make_uint4(v.x + 1u, (unsigned)n, acc, v.w + 3u);
    }
    __syncthreads();
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

__device__ __noinline__ unsigned touch(unsigned* a, int j, unsigned v) {   // the array's address escapes: it lives in the private segment (scratch memory)
  const unsigned r = a[j];
  a[(j + 5) & 15] = r + v;
  return r;
}

__global__ void __launch_bounds__(256) k_private(int passes, unsigned* sink, const int* perm) {
  unsigned a[16];
  for (int k = 0; k < 16; k++) a[k] = threadIdx.x + k;
  unsigned acc = 0;
  for (int n = 0; n < passes; n++) {
    const int j = perm[n & 15];                          // run-time index: the array lives in scratch memory
    acc += touch(a, j, acc + n);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const int WG = 768;
  uint4* d; unsigned* sink; int* perm;
  const size_t maxr = (size_t)512 << 10;
  (void)hipMalloc(&d, maxr * WG); (void)hipMalloc(&sink, 64); (void)hipMalloc(&perm, 64);
  (void)hipMemset(d, 0, maxr * WG);
  int hp[16]; for (int k = 0; k < 16; k++) hp[k] = (k * 7) & 15;
  (void)hipMemcpy(perm, hp, 64, hipMemcpyHostToDevice);
  (void)hipDeviceSynchronize();
  const size_t R[3] = {(size_t)16 << 10, (size_t)64 << 10, (size_t)512 << 10};
  const int N[3] = {1000, 250, 32};
  for (int c = 0; c < 3; c++) {
    hipLaunchKernelGGL(k_rewrite, dim3(WG), dim3(256), 0, 0, d, R[c] / 16, N[c], sink);
    (void)hipDeviceSynchronize();
    printf("k_rewrite launch %d: %zu KiB per workgroup x %d workgroups = %zu KiB working set, %d passes: %zu KiB stored and as many loaded\n", c, R[c] >> 10, WG,
           (R[c] * WG) >> 10, N[c], (R[c] * WG * N[c]) >> 10);
  }
  hipLaunchKernelGGL(k_private, dim3(WG), dim3(256), 0, 0, 100000, sink, perm);
  (void)hipDeviceSynchronize();
  printf("k_private: 64 B per lane x %d lanes = %d KiB of private segment in use, 100000 passes of one 4-byte load + one 4-byte store per lane: %zu KiB stored and as many loaded\n",
         WG * 256, WG * 256 * 64 >> 10, ((size_t)WG * 256 * 4 * 100000) >> 10);
  return 0;
}
