// thor_hip_wide.cpp - the engine sources (tk_*.h) compiled a THIRD time, in namespace tk_wide: workgroups of EIGHT wavefronts per superblock (512 threads,
// 256 VGPRs, 150 KB of LDS: one workgroup per CU).  A block decision's work queue holds ~44 items (skip / merge trials, one motion-search item per
// reference, 20 intra trials, 16 inter trials); eight wavefronts drain it in fewer rounds than four, the bi-prediction phase splits its rows eight ways.
// Slower per CU when the chip can be filled with four-wave workgroups, faster per superblock when it cannot: the backend (thor_hip.cpp:run_superblocks)
// launches this kernel when all streams of a run together never offer more superblocks than there are CUs (3840x2160: up to 17 streams) - the
// single-stream / drop-in operating point (round 6, call 11: one 3840x2160 stream 2.32 -> 2.86 Mpixels/s).  Same sources, same results: the parity suite
// runs its 8-bit goldens through all three kernels (THOR_HIP_KERNEL=std|lat|wide forces one).
#define TK_OCC 2
#define TK_WAVES 8
#define tk tk_wide
#define TK_ALT(name) thor_wide_##name
#include "tk_altbuild.h"
