// thor_hip_wide.cpp - the engine sources (tk_*.h) compiled a THIRD time, in namespace tk_wide: workgroups of EIGHT wavefronts per superblock (512 threads,
// 256 VGPRs, 150 KB of LDS: one workgroup per CU).  A block decision's work queue holds ~44 items (skip / merge trials, one motion-search item per
// reference, 20 intra trials, 16 inter trials); eight wavefronts drain it in fewer rounds than four, the bi-prediction phase splits its rows eight ways.
// It saturates at ~3/4 of the throughput build's peak (one workgroup per CU), which that build only reaches with well over a hundred streams, and finishes a
// superblock ~25 % sooner: the backend (thor_hip.cpp:run_superblocks) launches this kernel while all streams of a run together offer at most 2.5 superblocks
// per CU (3840x2160: up to 42 streams, 1920x1080: up to 80) - the single-stream / drop-in / few-stream operating points (round 6, calls 11-13: one
// 3840x2160 stream 2.32 -> 2.86 Mpixels/s, 32 streams 64.1 -> 80.2, 48 streams 91.5 -> 96.3).  Same sources, same results: the parity suite
// runs its 8-bit goldens through all three kernels (THOR_HIP_KERNEL=std|lat|wide forces one).
#define TK_OCC 2
#define TK_WAVES 8
#define tk tk_wide
#define TK_ALT(name) thor_wide_##name
#include "tk_altbuild.h"
