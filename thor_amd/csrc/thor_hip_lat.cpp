// thor_hip_lat.cpp - the engine sources (tk_*.h) compiled a SECOND time, in namespace tk_lat, with the register budget of two wavefronts per SIMD:
// k_superblocks<uint8_t> with 256 VGPRs (no register-pressure spills; the throughput build in thor_hip.cpp has 168 and spills) and 80 KB of LDS per
// workgroup (search windows for PUs up to 64x64).  Two such workgroups fit a CU instead of three - slower when the chip is full (round 4: -12 %), faster
// per superblock when it is not: the backend (thor_hip.cpp:run_superblocks) launches this kernel when a run cannot occupy more than two workgroups per CU
// anyway - few streams, the single-stream / drop-in operating point (round 6: 2.15 -> 2.33 Mpixels/s for one 3840x2160 stream).  Same sources, same
// results: the parity suite runs its small goldens through both kernels (THOR_HIP_KERNEL=std|lat forces one).
#define TK_OCC 2
#define tk tk_lat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "tk_kernel.h"

namespace tk {
__device__ Tables g_tab;
}

// (internal to libthor_hip.so: hidden symbols, called only by thor_hip.cpp)
#define TK_INTERNAL __attribute__((visibility("hidden")))
extern "C" {
// the constant tables of this copy of the engine (tk_tables.h: filled by the host side of thor_hip.cpp, identical bytes)
TK_INTERNAL int thor_lat_upload_tables(const void* tables, size_t bytes) {
  if (bytes != sizeof(tk::Tables)) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(tk::g_tab), tables, bytes) == hipSuccess ? 0 : 2;
}
// workgroups of the latency kernel that fit one CU
TK_INTERNAL int thor_lat_workgroups_per_cu(void) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tk::k_superblocks<uint8_t>, tk::kWgThreads, 0) != hipSuccess) return 0;
  return per_cu;
}
TK_INTERNAL int thor_lat_kernel_info(int* num_regs, int* lds_bytes, int* private_bytes) {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&tk::k_superblocks<uint8_t>)) != hipSuccess) return 1;
  if (num_regs) *num_regs = a.numRegs;
  if (lds_bytes) *lds_bytes = (int)a.sharedSizeBytes;
  if (private_bytes) *private_bytes = (int)a.localSizeBytes;
  return 0;
}
// launch: `jobs` = device array of FrameJob<uint8_t>, `dfargs` = the scheduler arguments (tk::DfArgs of thor_hip.cpp: same header, same layout)
TK_INTERNAL int thor_lat_launch_u8(int wgs, void* stream, const void* jobs, const void* dfargs, size_t dfargs_bytes, size_t job_bytes, size_t slot_bytes) {
  if (dfargs_bytes != sizeof(tk::DfArgs) || job_bytes != sizeof(tk::FrameJob<uint8_t>) || slot_bytes < sizeof(tk::BigWs<uint8_t>)) {
    fprintf(stderr, "thor_hip: the two builds of the engine disagree about a structure layout\n");
    return 1;
  }
  tk::DfArgs A;
  __builtin_memcpy(&A, dfargs, sizeof(A));
  hipLaunchKernelGGL(tk::k_superblocks<uint8_t>, dim3(wgs), dim3(tk::kWgThreads), 0, (hipStream_t)stream, (const tk::FrameJob<uint8_t>*)jobs, A);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
}
