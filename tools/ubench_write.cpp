// ubench_write.cpp - calibration of rocprofv3's WRITE_SIZE (and of FETCH_SIZE under stores) on gfx950 for the store patterns of this
// library: the guide (MI355X_MICROARCH.md, HBM section) calibrates only FETCH_SIZE on wide coalesced reads and calls WRITE_SIZE
// uncalibrated.  Every kernel writes each byte of a 1 GiB buffer (4x the Infinity Cache) exactly once:
//   k_store16 : 16 bytes per lane, fully coalesced (1 KiB per wave instruction)
//   k_store4  : one dword per lane (256 bytes per wave instruction)
//   k_store1  : ONE BYTE per lane (64 contiguous bytes per wave instruction: the sample-wise loops of the trial buffers)
//   k_rows1   : one byte per lane, 64-byte rows 4160 bytes apart written row by row (a 64-wide block copy into a frame plane)
// Run under `rocprofv3 --pmc WRITE_SIZE --kernel-trace` and `--pmc FETCH_SIZE --kernel-trace` and compare the counters (KiB) per
// kernel with the byte counts printed here.
//   hipcc --offload-arch=gfx950 -O2 -o ubench_write tools/ubench_write.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k_store16(uint4* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}
__global__ void k_store4(uint32_t* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
__global__ void k_store1(uint8_t* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint8_t)i;
}
__global__ void k_rows1(uint8_t* p, size_t rows, size_t pitch) {   // one wavefront per 64-byte row segment
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int lane = threadIdx.x & 63;
  for (size_t r = wave; r < rows; r += nw) p[r * pitch + 64 * (r & 31) + lane] = (uint8_t)r;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  uint8_t* d;
  hipMalloc(&d, bytes + 8192);
  hipMemset(d, 1, bytes + 8192);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k_store16, dim3(4096), dim3(256), 0, 0, (uint4*)d, bytes / 16);
  hipDeviceSynchronize();
  printf("k_store16: %zu bytes written = %zu KiB\n", bytes, bytes >> 10);
  hipLaunchKernelGGL(k_store4, dim3(4096), dim3(256), 0, 0, (uint32_t*)d, bytes / 4);
  hipDeviceSynchronize();
  printf("k_store4 : %zu bytes written = %zu KiB\n", bytes, bytes >> 10);
  hipLaunchKernelGGL(k_store1, dim3(4096), dim3(256), 0, 0, d, bytes);
  hipDeviceSynchronize();
  printf("k_store1 : %zu bytes written = %zu KiB\n", bytes, bytes >> 10);
  const size_t pitch = 4160, rows = bytes / pitch;
  hipLaunchKernelGGL(k_rows1, dim3(4096), dim3(256), 0, 0, d, rows, pitch);
  hipDeviceSynchronize();
  printf("k_rows1  : %zu bytes written (64 per row), %zu rows = %zu KiB; %zu KiB of 64-byte pieces, %zu KiB of 128-byte lines touched\n", rows * 64, rows, rows * 64 >> 10,
         rows * 64 >> 10, rows * 128 >> 10);
  return 0;
}
