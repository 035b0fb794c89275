// sched_stress.cpp - test infrastructure: the ready-task queues of the persistent superblock kernel (thor_amd/csrc/tk_sched.h)
// driven by OS threads instead of workgroups.  Every "workgroup" takes tasks with df_next, checks that the task's dependencies
// have finished, combines their payloads into its own (plain loads / stores: ThreadSanitizer sees a missing release/acquire),
// burns a pseudo-random amount of time and releases the successors with df_finish.
//   sched_stress S ROWS COLS WORKERS SEED   -> exit code 0 and a one-line summary, or a message and exit code 1
#define THOR_SCHED_HOSTTEST 1
#include "../../thor_amd/csrc/tk_sched.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
using namespace tk;

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: sched_stress S ROWS COLS WORKERS SEED [STAGGER]\n"); return 2; }
  const int S = atoi(argv[1]), rows = atoi(argv[2]), cols = atoi(argv[3]), W = atoi(argv[4]);
  const unsigned seed = (unsigned)atoi(argv[5]);
  const int stagger = argc > 6 ? atoi(argv[6]) : 0;
  const int nsb = rows * cols;
  const unsigned all = (unsigned)S * (unsigned)nsb;
  std::vector<unsigned> state(all, 0u);            // 0 not started, 1 running, 2 done (atomic)
  std::vector<unsigned long long> payload(all, 0);  // plain memory: ordered only by the scheduler's release/acquire
  // One launch: every stream's superblocks on the anti-diagonals [lo, hi) of `ranges` (nullptr: whole frames), as run_superblocks sets it up.
  auto launch = [&](const unsigned* ranges) -> int {
    std::vector<unsigned> queue(all, kDfEmpty), cnt(all, 0u);
    unsigned total = 0, n0 = 0;
    for (int s = 0; s < S; s++) {
      const int lo = ranges ? (int)(ranges[s] & 0xffffu) : 0, hi = ranges ? (int)(ranges[s] >> 16) : 0x7fff;
      for (int k = 0; k < rows; k++)
        for (int l = 0; l < cols; l++) {
          const int t = df_diag(k, l);
          if (t < lo || t >= hi) continue;
          total++;
          if (df_need(k, l, cols, lo) == 0) queue[n0++] = (unsigned)s * (unsigned)nsb + (unsigned)(k * cols + l);
        }
    }
    if (!total) return 0;
    DfCtl ctl = {0u, n0, 0u, 0u};
    DfArgs A;
    memset(&A, 0, sizeof(A));
    A.ctl = &ctl; A.queue = queue.data(); A.cnt = cnt.data(); A.S = S; A.nsb = nsb; A.cols = cols; A.rows = rows;
    A.spin_limit = 60ull * 100000000ull;
    A.range = ranges; A.total = total;
    std::vector<int> bad(W, 0);
    std::vector<std::thread> th;
    for (int w = 0; w < W; w++)
      th.emplace_back([&, w]() {
        unsigned rng = seed * 2654435761u + (unsigned)w * 40503u + 1u;
        for (;;) {
          const unsigned task = df_next(A, total);
          if (task == kDfEmpty) break;
          const int s = (int)(task / (unsigned)nsb), sb = (int)(task % (unsigned)nsb), k = sb / cols, l = sb % cols;
          if (__atomic_exchange_n(&state[task], 1u, __ATOMIC_ACQ_REL) != 0u) { bad[w] = 1; break; }   // handed out twice
          unsigned long long v = 1;
          const unsigned base = (unsigned)s * (unsigned)nsb;
          if (l > 0) {
            const unsigned d = base + (unsigned)(k * cols + l - 1);
            if (__atomic_load_n(&state[d], __ATOMIC_ACQUIRE) != 2u) { bad[w] = 2; break; }           // released before its dependency finished
            v += payload[d];
          }
          if (k > 0) {
            const int lu = l + 1 < cols ? l + 1 : cols - 1;
            const unsigned d = base + (unsigned)((k - 1) * cols + lu);
            if (__atomic_load_n(&state[d], __ATOMIC_ACQUIRE) != 2u) { bad[w] = 2; break; }
            v += payload[d];
          }
          rng = rng * 1664525u + 1013904223u;
          for (volatile unsigned spin = 0; spin < (rng >> 22); spin++) {}
          if ((rng & 7u) == 0) sched_yield();
          payload[task] = v;
          __atomic_store_n(&state[task], 2u, __ATOMIC_RELEASE);
          df_finish(A, s, k, l);
        }
      });
    for (auto& t : th) t.join();
    for (int w = 0; w < W; w++)
      if (bad[w]) { fprintf(stderr, "worker %d: %s\n", w, bad[w] == 1 ? "task handed out twice" : "task released before a dependency finished"); return 1; }
    if (ctl.error) { fprintf(stderr, "scheduler reported an error\n"); return 1; }
    if (ctl.tail != total) { fprintf(stderr, "%u of %u tasks handed out\n", ctl.tail, total); return 1; }
    return 0;
  };
  if (!stagger) {
    if (launch(nullptr)) return 1;
  } else {
    // the two-group schedule of Engine::encode_run for one frame: group 0 half a frame ahead of group 1
    const int T = (cols - 1) + 2 * (rows - 1) + 1, Th = (T + 1) / 2, S0 = S / 2;
    std::vector<unsigned> r(S);
    auto set = [&](int lo0, int hi0, int lo1, int hi1) { for (int s = 0; s < S; s++) r[s] = s < S0 ? ((unsigned)lo0 | ((unsigned)hi0 << 16)) : ((unsigned)lo1 | ((unsigned)hi1 << 16)); };
    set(0, Th, 0, 0);   if (launch(r.data())) return 1;
    set(Th, T, 0, Th);  if (launch(r.data())) return 1;
    set(0, 0, Th, T);   if (launch(r.data())) return 1;
  }
  const unsigned total = all;
  // payload of SB(k,l) = number of dependency paths into it + 1 ... same for every stream: compare against stream 0 computed serially
  std::vector<unsigned long long> ref(nsb, 0);
  for (int k = 0; k < rows; k++)
    for (int l = 0; l < cols; l++) {
      unsigned long long v = 1;
      if (l > 0) v += ref[k * cols + l - 1];
      if (k > 0) v += ref[(k - 1) * cols + (l + 1 < cols ? l + 1 : cols - 1)];
      ref[k * cols + l] = v;
    }
  for (unsigned t = 0; t < total; t++) {
    if (state[t] != 2u) { fprintf(stderr, "task %u not finished\n", t); return 1; }
    if (payload[t] != ref[t % (unsigned)nsb]) { fprintf(stderr, "task %u: payload %llu, expected %llu\n", t, payload[t], ref[t % (unsigned)nsb]); return 1; }
  }
  printf("ok: %u tasks, %d workers\n", total, W);
  return 0;
}
