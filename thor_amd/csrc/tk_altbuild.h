// tk_altbuild.h - body of an ALTERNATIVE BUILD of the engine sources (tk_*.h): thor_hip_lat.cpp and thor_hip_wide.cpp define the build's parameters
// (TK_OCC, TK_WAVES), re-name namespace tk (#define tk tk_xxx) and the prefix of the entry points (TK_ALT(name)) and include this file.  Each build is its
// own translation unit with its own copy of the constant tables; thor_hip.cpp (the throughput build + the whole host side) calls the hidden entry points.
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
TK_INTERNAL int TK_ALT(upload_tables)(const void* tables, size_t bytes) {
  if (bytes != sizeof(tk::Tables)) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(tk::g_tab), tables, bytes) == hipSuccess ? 0 : 2;
}
// workgroups of this build's 8-bit kernel that fit one CU, and wavefronts per workgroup (= BigWs slots the pool needs per workgroup)
TK_INTERNAL int TK_ALT(workgroups_per_cu)(void) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tk::k_superblocks<uint8_t>, tk::kWgThreads, 0) != hipSuccess) return 0;
  return per_cu;
}
TK_INTERNAL int TK_ALT(waves)(void) { return (int)tk::kWaves; }
TK_INTERNAL int TK_ALT(kernel_info)(int* num_regs, int* lds_bytes, int* private_bytes) {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&tk::k_superblocks<uint8_t>)) != hipSuccess) return 1;
  if (num_regs) *num_regs = a.numRegs;
  if (lds_bytes) *lds_bytes = (int)a.sharedSizeBytes;
  if (private_bytes) *private_bytes = (int)a.localSizeBytes;
  return 0;
}
// launch: `jobs` = device array of FrameJob<uint8_t>, `dfargs` = the scheduler arguments (tk::DfArgs of thor_hip.cpp: same header, same layout)
TK_INTERNAL int TK_ALT(launch_u8)(int wgs, void* stream, const void* jobs, const void* dfargs, size_t dfargs_bytes, size_t job_bytes, size_t slot_bytes) {
  if (dfargs_bytes != sizeof(tk::DfArgs) || job_bytes != sizeof(tk::FrameJob<uint8_t>) || slot_bytes < sizeof(tk::BigWs<uint8_t>)) {
    fprintf(stderr, "thor_hip: the builds of the engine disagree about a structure layout\n");
    return 1;
  }
  tk::DfArgs A;
  __builtin_memcpy(&A, dfargs, sizeof(A));
  hipLaunchKernelGGL(tk::k_superblocks<uint8_t>, dim3(wgs), dim3(tk::kWgThreads), 0, (hipStream_t)stream, (const tk::FrameJob<uint8_t>*)jobs, A);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
}
