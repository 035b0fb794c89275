// ubench_fetch.cpp - calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of this library
// ("calibrate FETCH_SIZE on a 4-byte-gather microkernel before applying the x2").
// Three kernels over a 1 GiB buffer (4x the Infinity Cache), each touching every 128-byte line exactly once:
//   k_stream16 : 16 bytes per lane, fully coalesced        (the pattern the guide's "x2" correction was measured on)
//   k_gather4  : ONE dword per 128-byte line and lane        (64 lines per wave-instruction: the motion search's gathers)
//   k_rows16   : 16 unaligned bytes per lane, lanes 4160 bytes apart (row segments of a padded 3840-wide plane)
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and compare FETCH_SIZE (KiB) per kernel with the bytes printed here.
//   hipcc --offload-arch=gfx950 -O2 -o ubench_fetch tools/ubench_fetch.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k_stream16(const uint4* p, size_t n, unsigned* out) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_gather4(const uint32_t* p, size_t lines, unsigned* out) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < lines; i += (size_t)gridDim.x * blockDim.x) acc += p[i * 32 + (i & 31)];
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_rows16(const uint8_t* p, size_t rows, size_t pitch, unsigned* out) {
  typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u4u;
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x) {
    const u4u v = *(const u4u*)(p + i * pitch + 37 + (i & 63));   // unaligned, inside one 128-byte line or straddling two
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  uint8_t* d;
  unsigned* out;
  hipMalloc(&d, bytes + 4096);
  hipMalloc(&out, 4);
  hipMemset(d, 1, bytes + 4096);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k_stream16, dim3(4096), dim3(256), 0, 0, (const uint4*)d, bytes / 16, out);
  hipDeviceSynchronize();
  printf("k_stream16: %zu bytes requested = %zu KiB, every 128-byte line once\n", bytes, bytes >> 10);
  hipLaunchKernelGGL(k_gather4, dim3(4096), dim3(256), 0, 0, (const uint32_t*)d, bytes / 128, out);
  hipDeviceSynchronize();
  printf("k_gather4 : %zu bytes requested (4 per line), %zu lines = %zu KiB of 128-byte lines touched\n", bytes / 32, bytes / 128, bytes >> 10);
  const size_t pitch = 4160, rows = bytes / pitch;
  hipLaunchKernelGGL(k_rows16, dim3(4096), dim3(256), 0, 0, (const uint8_t*)d, rows, pitch, out);
  hipDeviceSynchronize();
  printf("k_rows16  : %zu bytes requested (16 per row), %zu rows, 1-2 lines of 128 bytes each = %zu..%zu KiB touched\n", rows * 16, rows, rows * 128 >> 10, rows * 256 >> 10);
  return 0;
}
