// tk_kernel.h - the persistent, dependency-driven superblock kernel (one workgroup of kWaves wavefronts per superblock in flight; ready-task queue in
// tk_sched.h).  Its own header since round 6: the engine sources are compiled twice - thor_hip.cpp with the register budget of three wavefronts per SIMD
// (168 VGPRs, the throughput operating point: 768 resident workgroups) and thor_hip_lat.cpp with that of two (256 VGPRs: no register-pressure spills,
// 80 KB of LDS per workgroup = search windows for PUs up to 64x64), which the backend launches when a run cannot fill more than two workgroups per CU
// anyway (few streams: the single-stream / drop-in operating point).
#pragma once
#include "tk_block.h"
#include "tk_sched.h"

namespace tk {
// Workgroup = kWaves wavefronts on one superblock: wave 0 walks the quadtree (process_sb), the others are parked on
// the workgroup barrier and take work items of the block decisions (tk_block.h:mode_decision_par).  3 workgroups of
// 4 waves per CU = 3 waves per SIMD (168 VGPRs each), 768 workgroups resident on the chip.
enum { kWgThreads = 64 * kWaves };
enum { kOcc = TK_OCC };   // wavefronts per SIMD the register allocation is sized for (168 VGPRs); 2 and 4 measured slower (profiles/r02_ab_variants.md)
// 16-bit samples: two waves per SIMD (256 VGPRs; the 16-bit instances need ~245 and their 80 KB of LDS per workgroup allow two per CU anyway) -
// the register budget of a kernel is also the budget of every function only it calls.
template <typename PIX> __global__ __launch_bounds__(kWgThreads, (sizeof(PIX) == 1 ? (int)kOcc : 2)) void k_superblocks(const FrameJob<PIX>* jobs, DfArgs A) {
  __shared__ FrameJob<PIX> sJ;
  __shared__ WgShared sh;
  __shared__ SmallWs<PIX> sws[kWaves];
  __shared__ unsigned s_task;
  JobR<PIX> J = *ldsc(&sJ);
  const unsigned total = A.total;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
#ifdef THOR_PROF
  const Wg wg{wave, kWaves, sws[wave].prof};
#else
  const Wg wg{wave, kWaves};
#endif
  __shared__ TeamWs<PIX> s_view[kWaves];   // per-wave view of the workspaces: in LDS so that the callees read it with ds_read
  lds_st(&s_view[wave], make_ws(&sws[wave], &sh, (BigWs<PIX>*)(A.pool + ((size_t)blockIdx.x * kWaves + wave) * A.slot_bytes)));
  WsP<PIX> ws = ldsc(&s_view[wave]);
  const Team t = mk_team(lane, 64, sh.tabs.izz);
  xform_tables_fill(&sh.tabs, (int)threadIdx.x, kWgThreads);  // constant: once per workgroup
  for (;;) {
    __syncthreads();
    unsigned long long tpop = 0;
    if (threadIdx.x == 0) {
      if (A.times) tpop = wall_clock64();
      s_task = df_next(A, total);
    }
    __syncthreads();
    const unsigned task = (unsigned)__builtin_amdgcn_readfirstlane((int)s_task);
    if (task == kDfEmpty) break;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int sidx = (int)(task / (unsigned)A.nsb), sb = (int)(task % (unsigned)A.nsb);
    const int k = sb / A.cols, l = sb % A.cols;
    {
      const uint32_t* src = (const uint32_t*)&jobs[sidx];
      uint32_t* dst = (uint32_t*)&sJ;
      for (int i = threadIdx.x; i < (int)(sizeof(FrameJob<PIX>) / 4); i += kWgThreads) dst[i] = src[i];
    }
#ifdef THOR_PROF
    if (lane < kProfSlots) sws[wave].prof[lane] = 0;
#endif
    __syncthreads();
    if (wave == 0) {
      if (A.times && lane == 0) { A.times[3 * (size_t)task] = tpop; A.times[3 * (size_t)task + 1] = wall_clock64(); }
      BitSink out;
      out.buf = J.sb_bits + (size_t)sb * J.sb_words;
      out.pos = 0;
      out.cap = J.sb_words * 32;
      out.emit = 1;
      out.ovf = 0;
      process_sb(wg, t, J, ws, k * kMaxSb, l * kMaxSb, out);
      if (lane == 0) {
        J.sb_nbits[sb] = out.pos;
        J.sb_status[sb] = out.ovf;
      }
    } else
      wg_helper_loop(wg, t, J, ws);
#ifdef THOR_PROF
    __syncthreads();
    if (J.prof && lane < kProfSlots) atomicAdd((unsigned long long*)&J.prof[lane], (unsigned long long)sws[wave].prof[lane]);
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (A.times) A.times[3 * (size_t)task + 2] = wall_clock64();
      df_finish(A, sidx, k, l);
    }
  }
}

}  // namespace tk
