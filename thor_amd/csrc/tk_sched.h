// tk_sched.h - ready-task queues of the persistent superblock kernel (thor_hip.cpp:k_superblocks).
//
// SB(k,l) of a stream needs SB(k,l-1) and SB(k-1,l+1) of the same stream (enc/encode_frame.c walks the superblocks in raster
// order; the dependencies are those of the motion-vector / intra / context neighbours).  One launch per
// frame covers every superblock of every stream; resident workgroups take ready (stream, SB) tasks until all are taken.  A
// finishing task bumps the dependency counters of its successors and pushes those that became ready.  Every task is pushed
// exactly once, a workgroup only ever holds a task that is ready and running tasks never wait: no deadlock whatever the
// dispatch order or residency.  Publication uses agent-scope release (push) / acquire (after the pop), which also orders the
// data across the 8 XCD L2s.
//
// One queue discipline: ready-order FIFO over the `queue` array (one slot per task of the frame, single use, no wrap-around) - pushes
// take a tail ticket, idle workgroups take a head ticket and wait for that slot to be filled.  (A second discipline, "laggards
// first" - one FIFO per anti-diagonal of the superblock grid, lowest first - was built at the end of round 3 and measured in round
// 4: bit-exact, and no faster at 3840x2160 x 128 streams (114.99 vs 114.84 Mpx/s, 88 % of the workgroup-time busy either way,
// profiles/r04_call2_ab.md); it was removed again.)
//
// The functions below are written against a small set of atomics macros so that tests/hostsim/sched_stress.cpp can run the
// very same protocol with OS threads (THOR_SCHED_HOSTTEST); the device build maps them to agent-scope HIP atomics.
//
// Partial frames (round 5): a launch may cover, per stream, only the superblocks of a range [lo, hi) of anti-diagonals t = l + 2k (both
// dependencies of a superblock lie on smaller anti-diagonals, so every prefix of the order is closed under the dependencies).  The host
// codes the streams in two groups half a frame apart (tk_encoder.h:encode_run): one launch carries the second half of one group's frame -
// its narrowing end - and the first half of the other group's, so the ramp-up of one group fills the slots the ramp-down of the other
// leaves idle.  A dependency below `lo` finished in an earlier launch and is not counted; a successor at or beyond `hi` is left to a
// later launch, which starts with every superblock whose dependencies all lie below its `lo`.
#pragma once
#include <stdint.h>

#if defined(THOR_SCHED_HOSTTEST)
#include <sched.h>
#include <time.h>
#define DF_FN static inline
#define DF_HD static inline
// (acquire loads instead of relaxed loads + the fence the kernel executes after the pop: ThreadSanitizer does not model fences)
#define DF_LOAD(p) __atomic_load_n((p), __ATOMIC_ACQUIRE)
#define DF_ADD(p, v) __atomic_fetch_add((p), (v), __ATOMIC_RELAXED)
#define DF_ADD_ACQ_REL(p, v) __atomic_fetch_add((p), (v), __ATOMIC_ACQ_REL)
#define DF_STORE_RELEASE(p, v) __atomic_store_n((p), (v), __ATOMIC_RELEASE)
#define DF_EXCHANGE(p, v) __atomic_exchange_n((p), (v), __ATOMIC_RELAXED)
#define DF_BACKOFF() sched_yield()
static inline unsigned long long df_clock() {  // 100 MHz ticks like the device's wall clock
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (unsigned long long)ts.tv_sec * 100000000ull + (unsigned long long)ts.tv_nsec / 10ull;
}
#else
#define DF_FN __device__ inline
#define DF_HD __host__ __device__ inline
#define DF_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define DF_ADD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define DF_ADD_ACQ_REL(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
#define DF_STORE_RELEASE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)
#define DF_EXCHANGE(p, v) __hip_atomic_exchange((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// ~14 us between polls: an idle workgroup must not compete with the working ones for the L2
#define DF_BACKOFF() do { for (int i_ = 0; i_ < 4; i_++) __builtin_amdgcn_s_sleep(127); } while (0)
__device__ inline unsigned long long df_clock() { return wall_clock64(); }
#endif

namespace tk {

struct DfCtl {
  unsigned head, tail;   // pop / push tickets
  unsigned error;
  unsigned pad;
};
struct DfArgs {
  DfCtl* ctl;
  unsigned* queue;             // [S*nsb] task ids (stream*nsb + sb), 0xffffffff = not filled yet
  unsigned* cnt;               // [S*nsb] finished-dependency counters
  uint8_t* pool;               // one BigWs scratch slot per wavefront
  size_t slot_bytes;
  unsigned long long* times;   // optional [S*nsb][3] pop/start/end wall clock (100 MHz)
  int S, nsb, cols, rows;
  unsigned long long spin_limit;  // wall-clock ticks a workgroup may wait
  const unsigned* range;          // [S] anti-diagonals [lo, hi) of every stream in this launch, lo | hi << 16; nullptr: whole frames
  unsigned total;                 // tasks of this launch
};
// anti-diagonal of SB(k,l) and of its two dependencies: left (k,l-1) and up-right (k-1,l+1) - (k-1,l) in the last column
DF_HD int df_diag(int k, int l) { return l + 2 * k; }
DF_HD int df_need(int k, int l, int cols, int lo) {
  const int t = l + 2 * k;
  return (l > 0 && t - 1 >= lo ? 1 : 0) + (k > 0 && (l == cols - 1 ? t - 2 : t - 1) >= lo ? 1 : 0);
}
static const unsigned kDfEmpty = 0xffffffffu;

// Wait until `slot` holds a task id.  Polls with RELAXED agent-scope loads: an acquire load in the loop would issue a
// buffer_inv (a whole-L2 invalidate on this XCD) per poll and starve every working wavefront; the single acquire fence the
// caller executes afterwards orders the dependent reads.  kDfEmpty: another workgroup reported an error, or the wait limit.
DF_FN unsigned df_wait_slot(const DfArgs& A, unsigned slot) {
  const unsigned long long t0 = df_clock();
  unsigned n = 0, v;
  while ((v = DF_LOAD(&A.queue[slot])) == kDfEmpty) {
    DF_BACKOFF();
    if ((++n & 63u) == 0) {
      if (DF_LOAD(&A.ctl->error)) return kDfEmpty;
      if (df_clock() - t0 > A.spin_limit) { DF_EXCHANGE(&A.ctl->error, 1u); return kDfEmpty; }
    }
  }
  return v;
}

// Task `id` = SB(k,l) of some stream has become ready.
DF_FN void df_push(const DfArgs& A, unsigned id) {
  const unsigned p = DF_ADD(&A.ctl->tail, 1u);
  DF_STORE_RELEASE(&A.queue[p], id);
}

// A dependency of SB(k,l) of the stream whose tasks start at `base` has finished (lo, hi: the stream's range in this launch).
DF_FN void df_done_dep(const DfArgs& A, unsigned base, int k, int l, int lo, int hi) {
  if (df_diag(k, l) >= hi) return;   // not part of this launch: a later one starts it
  const unsigned id = base + (unsigned)(k * A.cols + l);
  const unsigned need = (unsigned)df_need(k, l, A.cols, lo);
  const unsigned old = DF_ADD_ACQ_REL(&A.cnt[id], 1u);
  if (old + 1 == need) df_push(A, id);
}

// SB(k,l) of stream `sidx` has finished: release its successors.
DF_FN void df_finish(const DfArgs& A, int sidx, int k, int l) {
  const unsigned base = (unsigned)sidx * (unsigned)A.nsb;
  int lo = 0, hi = 0x7fff;
  if (A.range) { const unsigned r = A.range[sidx]; lo = (int)(r & 0xffffu); hi = (int)(r >> 16); }
  if (l + 1 < A.cols) df_done_dep(A, base, k, l + 1, lo, hi);               // right neighbour: its left dependency
  if (k + 1 < A.rows) {
    if (l >= 1) df_done_dep(A, base, k + 1, l - 1, lo, hi);                 // down-left: its up-right dependency
    if (l == A.cols - 1) df_done_dep(A, base, k + 1, l, lo, hi);            // last column: the SB below uses (k,l) as "up-right"
  }
}

// Next task for an idle workgroup (called by ONE thread of it), kDfEmpty when every task of the frame has been handed out
// (or on error).
DF_FN unsigned df_next(const DfArgs& A, unsigned total) {
  const unsigned slot = DF_ADD(&A.ctl->head, 1u);
  return slot < total ? df_wait_slot(A, slot) : kDfEmpty;
}

}  // namespace tk
