// thor_hip_lat.cpp - the engine sources (tk_*.h) compiled a SECOND time, in namespace tk_lat, with the register budget of two wavefronts per SIMD:
// k_superblocks<uint8_t> with 256 VGPRs (no register-pressure spills; the throughput build in thor_hip.cpp has 168 and spills) and 80 KB of LDS per
// workgroup (search windows for PUs up to 64x64).  Two such workgroups fit a CU instead of three - slower when the chip is full (round 4: -12 %), faster
// per superblock when it is not (round 6: 2.15 -> 2.33 Mpixels/s for one 3840x2160 stream).  Since call 13 of round 6 the eight-wavefront build
// (thor_hip_wide.cpp) covers that whole range and beats this one by ~25 % there, so the backend no longer picks this kernel by itself: it runs when forced
// (THOR_HIP_KERNEL=lat) - the parity suite runs its 8-bit goldens through all three kernels - and stays as the A/B reference of the spill-free register budget.
#define TK_OCC 2
#define tk tk_lat
#define TK_ALT(name) thor_lat_##name
#include "tk_altbuild.h"
