// thor_amd: MI355X-native per-block encode path for the Thor codec.
// tk_common.h - build modes, team abstraction, basic codec types and constant tables.
//
// The engine sources (tk_*.h) are written once as "team-cooperative" code: every
// function is executed by all lanes of a team in lock step (uniform control
// flow), data-parallel loops are strided over team.rank/team.size and phases are
// separated by team.sync().  On the GPU a team is one 64-lane wavefront
// (gfx950, wave64) working on one superblock.  With -DTHOR_HOSTSIM the very same
// source compiles with g++ as a 1-lane team; that build exists ONLY so that the
// bit-exactness of the algorithm can be developed and regression-tested against
// the reference encoder in a container without a GPU (tests/ only - the product
// library contains no CPU path).
#pragma once
#include <stdint.h>
#include <stddef.h>
// Wave-uniform values (arguments of the hot callees, running minima, syntax parameters) are moved to scalar registers
// with readfirstlane (TKU / tk_uniform*): function arguments arrive in vector registers, and a branch on a vector register is
// exec-mask code.  The host simulation checks the uniformity assumption (tk_uniform aborts when lanes disagree).
// Wave reductions use DPP row operations + v_readlane instead of ds_bpermute shuffles (profiles/r03_call2_findings.md).
#if defined(THOR_HOSTSIM)
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <sched.h>
#define TK_DEV static inline
#define TK_HD static inline
#define TK_CONST static const
#define TK_HOST 1
#else
#include <hip/hip_runtime.h>
#define TK_DEV __device__ __forceinline__
#define TK_HD __host__ __device__ __forceinline__
#define TK_DEVNI __device__ __noinline__
#define TK_CONST __device__ const
#define TK_HOST 0
#endif
#if TK_HOST
#define TK_DEVNI static
#endif
// The fork/join functions of a block decision (tk_block.h: md_worker_sp, mode_decision_par) save and restore every callee-saved register of the
// 168-VGPR budget (64 VGPRs + 36 SGPRs) on every wave of every decision although their callers keep nothing in them.  -DTK_MDW_INLINE inlines
// them into the kernel: measured in round 6 (profiles/r06_traffic_attribution.md) - HBM-side traffic -6 % (3 of 20 store instructions per pixel
// gone), throughput -0.9 %: the scratch traffic is waste, not the limiter, and the calls stay.
#if defined(TK_MDW_INLINE)
#define TK_MDW TK_DEV
#else
#define TK_MDW TK_DEVNI
#endif

namespace tk {

#if TK_HOST && defined(THOR_HOSTSIM_LANES)
// Multi-lane host simulation (tests/hostsim built with -DTHOR_HOSTSIM_LANES): every lane of a team is an OS thread;
// the cross-lane primitives below are implemented with one exchange operation (every lane publishes a 64-bit value,
// all lanes read the published values) provided by tests/hostsim/hostsim.cpp.  Test infrastructure only.
namespace hostlanes {
void barrier();
const unsigned long long* exchange_begin(unsigned long long v);  // slots[lane] of every lane, valid until exchange_end()
void exchange_end();
int lanes();
int rank();
}  // namespace hostlanes
#define TK_LANES 1
#else
#define TK_LANES 0
#endif

// ---------------------------------------------------------------------------------
// Team: the cooperating lane group.
// ---------------------------------------------------------------------------------
struct Team {
  int rank;
#if TK_HOST
  int size;
#else
  // On the device a team is always one 64-lane wavefront: a compile-time constant, so loop strides, lane-group arithmetic and
  // the branches on them are immediates / scalar code instead of values that arrive in a vector register with every call.
  static constexpr int size = 64;
#endif
  // team-local copy of the scan-order tables (scan index -> position): [0,16) 4x4, [16,80) 8x8, [80,336) 16x16.
  // On the device it points into LDS (XformWs::izz) so the per-coefficient lookups of quantisation and bit
  // counting do not take a global-memory round trip each; unused (nullptr) on the host simulation.
  const int16_t* izz = nullptr;
  // sync(): every lane of the TEAM has finished its earlier LDS / scratch accesses before any lane continues.  On the
  // device a team is ONE wavefront, whose lanes run in lock step, so this is only a memory fence (outstanding LDS and
  // vector-memory operations complete; no s_barrier): the workgroup holds several wavefronts that execute different
  // code (Wg below) and a workgroup barrier here would dead-lock them.
  // block_sync(): barrier over a whole thread block for the few kernels that use a block-wide team (CDEF selection).
#if TK_LANES
  inline void sync() const { hostlanes::barrier(); }
  inline void block_sync() const { hostlanes::barrier(); }
#elif TK_HOST
  inline void sync() const {}
  inline void block_sync() const {}
#else
  __device__ __forceinline__ void sync() const {
    // (a wavefront-scope fence, with or without s_waitcnt lgkmcnt(0), measured the same as this within noise on the MI355X:
    // profiles/r03_call2_findings.md - the waits on outstanding stores are not what the waves spend their time on)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  __device__ __forceinline__ void block_sync() const { __syncthreads(); }
#endif
};

// mk_team(rank, size, izz): `size` must be 64 on the device.
TK_DEV Team mk_team(int rank, int size, const int16_t* izz = nullptr) {
  Team t;
  t.rank = rank;
#if TK_HOST
  t.size = size;
#else
  (void)size;
#endif
  t.izz = izz;
  return t;
}
#if !TK_HOST
// A whole thread block as one team (CDEF strength selection): block-strided loops separated by block barriers.
struct BlockTeam {
  int rank;
  int size;
  __device__ __forceinline__ void block_sync() const { __syncthreads(); }
};
#endif

// ---------------------------------------------------------------------------------
// Wg: the wavefronts of one workgroup that cooperate on one superblock.  Wave 0 (the "master") walks the quadtree;
// the other waves are parked on the workgroup barrier and are woken for the parallel regions of a block decision
// (tk_block.h:mode_decision_par).  The host simulation runs one OS thread per wave (-DTHOR_HOSTSIM_WAVES=N, 1-lane
// teams) or a single wave (everything else).
// ---------------------------------------------------------------------------------
#if !TK_HOST
#ifndef TK_WAVES
#define TK_WAVES 4
#endif
enum { kWaves = TK_WAVES };   // wavefronts per workgroup = per superblock in flight (A/B builds: -DTK_WAVES=1 / 2)
#elif defined(THOR_HOSTSIM_WAVES)
enum { kWaves = THOR_HOSTSIM_WAVES };
#else
enum { kWaves = 1 };
#endif
#if TK_HOST && defined(THOR_HOSTSIM_WAVES)
namespace hostwaves { void barrier(); }
#endif
struct Wg {
  int wave, nwaves;
#if !TK_HOST && defined(THOR_PROF)
  long long* prof;  // this wave's cycle counters: slot 26 = time spent in workgroup barriers
  __device__ __forceinline__ void barrier() const {
    const long long t0 = (long long)__builtin_readcyclecounter();
    __syncthreads();
    if ((threadIdx.x & 63) == 0) prof[26] += (long long)__builtin_readcyclecounter() - t0;
  }
#elif !TK_HOST
  __device__ __forceinline__ void barrier() const { __syncthreads(); }
#elif defined(THOR_HOSTSIM_WAVES)
  inline void barrier() const { hostwaves::barrier(); }
#else
  inline void barrier() const {}
#endif
};
// Atomics on workgroup-shared state (always LDS on the device).  Call from ONE lane of a wave.
TK_DEV int wg_fetch_add(int* p, int v) {
#if TK_HOST
  return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL);
#else
  return __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)p, v, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
TK_DEV void wg_min64(unsigned long long* p, unsigned long long v) {
#if TK_HOST
  unsigned long long cur = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v < cur && !__atomic_compare_exchange_n(p, &cur, v, true, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED)) {}
#else
  __hip_atomic_fetch_min((__attribute__((address_space(3))) unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
TK_DEV int wg_load_acquire(const int* p) {
#if TK_HOST
  return __atomic_load_n(p, __ATOMIC_ACQUIRE);
#else
  return __hip_atomic_load((const __attribute__((address_space(3))) int*)p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
TK_DEV void wg_store_release(int* p, int v) {
#if TK_HOST
  __atomic_store_n(p, v, __ATOMIC_RELEASE);
#else
  __hip_atomic_store((__attribute__((address_space(3))) int*)p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
// compare-and-swap; returns 1 when *p was `expected` and is now `desired`
TK_DEV int wg_cas(int* p, int expected, int desired) {
#if TK_HOST
  return __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE) ? 1 : 0;
#else
  return __hip_atomic_compare_exchange_strong((__attribute__((address_space(3))) int*)p, &expected, desired, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE,
                                              __HIP_MEMORY_SCOPE_WORKGROUP) ? 1 : 0;
#endif
}
// Back-off inside a wait on another wave of the workgroup.
TK_DEV void wg_pause() {
#if TK_HOST
  sched_yield();
#else
  __builtin_amdgcn_s_sleep(8);
#endif
}
// Wall clock for the limit of such a wait: 100 MHz ticks on the device (s_memrealtime) and on the host.
#if TK_HOST
}  // namespace tk
#include <time.h>
namespace tk {
#endif
// Limit of a wait on another wave of the workgroup, in 100 MHz ticks: 8 s on the device (a search item takes milliseconds).  The host
// simulation runs waves as OS threads - a sanitizer build or an oversubscribed machine can need far longer for one item without any protocol
// error - so its limit is 120 s, or THOR_HOSTSIM_WAIT_S seconds.
#if TK_HOST
static inline unsigned long long wg_wait_limit() {
  static const unsigned long long lim = [] { const char* e = getenv("THOR_HOSTSIM_WAIT_S"); return (unsigned long long)((e && atof(e) > 0 ? atof(e) : 120.0) * 1e8); }();   // thread-safe initialisation
  return lim;
}
#define kWgWaitLimit wg_wait_limit()
#else
enum { kWgWaitLimit = 800000000 };   // 8 s
#endif
TK_DEV unsigned long long wg_clock() {
#if TK_HOST
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (unsigned long long)ts.tv_sec * 100000000ull + (unsigned long long)ts.tv_nsec / 10ull;
#else
  return wall_clock64();
#endif
}
// A wait on another wave that cannot end (a protocol error): stop the kernel / the simulation loudly instead of spinning for ever.
TK_DEV void wg_wait_failed() {
#if TK_HOST
  fprintf(stderr, "thor: a wavefront waited for another one for ever\n");
  abort();
#else
  __builtin_trap();
#endif
}
TK_DEV unsigned long long wg_load64(const unsigned long long* p) {
#if TK_HOST
  return __atomic_load_n(p, __ATOMIC_RELAXED);
#else
  return __hip_atomic_load((const __attribute__((address_space(3))) unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}

TK_DEV void team_add(int* p, int v) {
#if TK_LANES
  __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);
#elif TK_HOST
  *p += v;
#else
  atomicAdd(p, v);
#endif
}
TK_DEV void team_add64(unsigned long long* p, unsigned long long v) {
#if TK_LANES
  __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);
#elif TK_HOST
  *p += v;
#else
  atomicAdd(p, v);
#endif
}
TK_DEV void team_or(unsigned* p, unsigned v) {
#if TK_LANES
  __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST);
#elif TK_HOST
  *p |= v;
#else
  atomicOr(p, v);
#endif
}

#if !TK_HOST
// Wave-level reductions without the LDS crossbar: DPP moves inside the 16-lane rows (xor 1, xor 2, then the half-row and
// row mirrors, which equal xor 4 / xor 8 once the lower levels are uniform), v_readlane across the four rows.  A
// ds_bpermute (what __shfl_xor compiles to) is an LDS-pipe round trip per step; a DPP operand is part of the VALU op.
// Full EXEC required (team-cooperative code runs its reductions in uniform control flow).
template <int CTRL> __device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
enum { DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140 };
__device__ __forceinline__ int row_sum_dpp(int v) {
  v += dpp_mov<DPP_XOR1>(v); v += dpp_mov<DPP_XOR2>(v); v += dpp_mov<DPP_HALF_MIRROR>(v); v += dpp_mov<DPP_MIRROR>(v);
  return v;
}
__device__ __forceinline__ int wave_sum_dpp(int v) {
  v = row_sum_dpp(v);
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ unsigned wave_min_u32_dpp(unsigned v) {
  auto mn = [](unsigned a, unsigned b) { return a < b ? a : b; };
  v = mn(v, (unsigned)dpp_mov<DPP_XOR1>((int)v)); v = mn(v, (unsigned)dpp_mov<DPP_XOR2>((int)v));
  v = mn(v, (unsigned)dpp_mov<DPP_HALF_MIRROR>((int)v)); v = mn(v, (unsigned)dpp_mov<DPP_MIRROR>((int)v));
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return mn(mn(a, b), mn(c, d));
}
__device__ __forceinline__ int wave_max_i32_dpp(int v) {
  auto mx = [](int a, int b) { return a > b ? a : b; };
  v = mx(v, dpp_mov<DPP_XOR1>(v)); v = mx(v, dpp_mov<DPP_XOR2>(v)); v = mx(v, dpp_mov<DPP_HALF_MIRROR>(v)); v = mx(v, dpp_mov<DPP_MIRROR>(v));
  return mx(mx(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), mx(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
template <int CTRL> __device__ __forceinline__ unsigned long long dpp_mov64(unsigned long long v) {
  const unsigned lo = (unsigned)dpp_mov<CTRL>((int)(unsigned)v), hi = (unsigned)dpp_mov<CTRL>((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_min64_dpp(unsigned long long v) {
  auto mn = [](unsigned long long a, unsigned long long b) { return a < b ? a : b; };
  v = mn(v, dpp_mov64<DPP_XOR1>(v)); v = mn(v, dpp_mov64<DPP_XOR2>(v)); v = mn(v, dpp_mov64<DPP_HALF_MIRROR>(v)); v = mn(v, dpp_mov64<DPP_MIRROR>(v));
  return mn(mn(readlane64(v, 0), readlane64(v, 16)), mn(readlane64(v, 32), readlane64(v, 48)));
}
__device__ __forceinline__ unsigned long long wave_sum64_dpp(unsigned long long v) {
  v += dpp_mov64<DPP_XOR1>(v); v += dpp_mov64<DPP_XOR2>(v); v += dpp_mov64<DPP_HALF_MIRROR>(v); v += dpp_mov64<DPP_MIRROR>(v);
  return readlane64(v, 0) + readlane64(v, 16) + readlane64(v, 32) + readlane64(v, 48);
}
#endif

// Cross-lane helpers.  A team of 1 lane (host simulation) degenerates to the identity, so code
// written against them is also the serial algorithm.
TK_DEV unsigned long long team_ballot(const Team t, int pred) {
#if TK_LANES
  const unsigned long long* g = hostlanes::exchange_begin(pred ? 1ull : 0ull);
  unsigned long long m = 0;
  for (int l = 0; l < t.size; l++) m |= g[l] << l;
  hostlanes::exchange_end();
  return m;
#elif TK_HOST
  (void)t;
  return pred ? 1ull : 0ull;
#else
  (void)t;
  return __ballot(pred);
#endif
}
TK_DEV int team_sum(const Team t, int v) {
#if TK_LANES
  const unsigned long long* g = hostlanes::exchange_begin((unsigned long long)(long long)v);
  int r = 0;
  for (int l = 0; l < t.size; l++) r += (int)(long long)g[l];
  hostlanes::exchange_end();
  return r;
#elif TK_HOST
  (void)t;
  return v;
#else
  (void)t;
  return wave_sum_dpp(v);
#endif
}
TK_DEV int team_max(const Team t, int v) {
#if TK_LANES
  const unsigned long long* g = hostlanes::exchange_begin((unsigned long long)(long long)v);
  int r = v;
  for (int l = 0; l < t.size; l++) r = (int)(long long)g[l] > r ? (int)(long long)g[l] : r;
  hostlanes::exchange_end();
  return r;
#elif TK_HOST
  (void)t;
  return v;
#else
  (void)t;
  return wave_max_i32_dpp(v);
#endif
}
TK_DEV int team_shfl_xor(const Team t, int v, int d) {
#if TK_LANES
  const unsigned long long* g = hostlanes::exchange_begin((unsigned long long)(long long)v);
  const int src = t.rank ^ d;
  const int r = src < t.size ? (int)(long long)g[src] : v;
  hostlanes::exchange_end();
  return r;
#elif TK_HOST
  (void)t; (void)d;
  return v;
#else
  (void)t;
  return __shfl_xor(v, d);
#endif
}
// sum over aligned groups of G lanes (G a power of two <= team size), result in every lane of the group
TK_DEV int team_group_sum(const Team t, int v, int G) {
#if !TK_HOST
  (void)t;
  if (G >= 2) v += dpp_mov<DPP_XOR1>(v);
  if (G >= 4) v += dpp_mov<DPP_XOR2>(v);
  if (G >= 8) v += dpp_mov<DPP_HALF_MIRROR>(v);
  if (G >= 16) v += dpp_mov<DPP_MIRROR>(v);
  if (G >= 32) v += __shfl_xor(v, 16);
  if (G >= 64) v += __shfl_xor(v, 32);
  return v;
#else
  for (int d = G >> 1; d >= 1; d >>= 1) v += team_shfl_xor(t, v, d);
  return v;
#endif
}
// value of lane `lane` (wave-uniform index) in every lane of the team
TK_DEV int team_read_lane(const Team t, int v, int lane) {
#if TK_LANES
  const unsigned long long* g = hostlanes::exchange_begin((unsigned long long)(long long)v);
  const int r = (int)(long long)g[lane < t.size ? lane : 0];
  hostlanes::exchange_end();
  return r;
#elif TK_HOST
  (void)t; (void)lane;
  return v;
#else
  (void)t;
  return __builtin_amdgcn_readlane(v, lane);
#endif
}
// minimum over the team, wave-uniform result
TK_DEV unsigned team_min32(const Team t, unsigned v) {
#if TK_LANES
  const unsigned long long* g = hostlanes::exchange_begin((unsigned long long)v);
  unsigned r = v;
  for (int l = 0; l < t.size; l++) r = (unsigned)g[l] < r ? (unsigned)g[l] : r;
  hostlanes::exchange_end();
  return r;
#elif TK_HOST
  (void)t;
  return v;
#else
  (void)t;
  return wave_min_u32_dpp(v);
#endif
}
TK_DEV unsigned long long team_min64(const Team t, unsigned long long v) {
#if TK_LANES
  const unsigned long long* g = hostlanes::exchange_begin(v);
  unsigned long long r = v;
  for (int l = 0; l < t.size; l++) r = g[l] < r ? g[l] : r;
  hostlanes::exchange_end();
  return r;
#elif TK_HOST
  (void)t;
  return v;
#else
  (void)t;
  return wave_min64_dpp(v);
#endif
}
// Value known to be identical in every lane of the team: on the device it is moved to a scalar register so that
// selects / branches on it are scalar instead of exec-mask juggling (the compiler cannot prove uniformity of values
// that went through memory or shuffles).
TK_DEV int tk_uniform(int v) {
#if TK_LANES
  const unsigned long long* g = hostlanes::exchange_begin((unsigned long long)(long long)v);
  for (int l = 0; l < hostlanes::lanes(); l++)
    if ((int)(long long)g[l] != v) { fprintf(stderr, "tk_uniform: value differs between lanes (%d vs %d)\n", v, (int)(long long)g[l]); abort(); }
  hostlanes::exchange_end();
  return v;
#elif TK_HOST
  return v;
#else
  return __builtin_amdgcn_readfirstlane(v);
#endif
}
// value of lane 0 in every lane of the team
TK_DEV int team_bcast0(const Team t, int v) {
#if TK_LANES
  const unsigned long long* g = hostlanes::exchange_begin((unsigned long long)(long long)v);
  const int r = (int)(long long)g[0];
  hostlanes::exchange_end();
  (void)t;
  return r;
#elif TK_HOST
  (void)t;
  return v;
#else
  (void)t;
  return __builtin_amdgcn_readfirstlane(v);
#endif
}
TK_DEV unsigned long long tk_uniform64(unsigned long long v) {
#if TK_LANES
  const unsigned long long* g = hostlanes::exchange_begin(v);
  for (int l = 0; l < hostlanes::lanes(); l++)
    if (g[l] != v) { fprintf(stderr, "tk_uniform64: value differs between lanes\n"); abort(); }
  hostlanes::exchange_end();
  return v;
#elif TK_HOST
  return v;
#else
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
#endif
}
TK_DEV double tk_uniform_f64(double d) {
  unsigned long long b;
  __builtin_memcpy(&b, &d, 8);
  b = tk_uniform64(b);
  __builtin_memcpy(&d, &b, 8);
  return d;
}
template <class T> TK_DEV T* tk_uniform_ptr(T* p) { return (T*)(uintptr_t)tk_uniform64((unsigned long long)(uintptr_t)p); }
#define TKU(x) tk_uniform(x)
#define TKU64(x) tk_uniform64(x)
// sum over the team, result in every lane (xor-shuffle butterfly; identity for a 1-lane team)
TK_DEV unsigned long long team_sum64(const Team t, unsigned long long v) {
#if TK_LANES
  const unsigned long long* g = hostlanes::exchange_begin(v);
  unsigned long long r = 0;
  for (int l = 0; l < t.size; l++) r += g[l];
  hostlanes::exchange_end();
  return r;
#elif TK_HOST
  (void)t;
  return v;
#else
  (void)t;
  return wave_sum64_dpp(v);
#endif
}
// index of the highest set bit of m strictly below `rank`, or -1
TK_DEV int prev_set(unsigned long long m, int rank) {
  m &= (rank >= 64) ? ~0ull : ((1ull << rank) - 1ull);
  if (!m) return -1;
#if TK_HOST
  return 63 - __builtin_clzll(m);
#else
  return 63 - __clzll((long long)m);
#endif
}
TK_DEV int top_set(unsigned long long m) { return prev_set(m, 64); }

// IEEE double multiply-add WITHOUT contraction: the reference is built -std=c99
// (=> -ffp-contract=off), so lambda*bits+0.5 is a rounded product followed by a
// rounded sum.
TK_DEV double mul_add_nofma(double a, double b, double c) {
#if TK_HOST
  volatile double p = a * b;
  return p + c;
#else
  return __dadd_rn(__dmul_rn(a, b), c);
#endif
}

// On the device the per-team small state (transform workspace, intra edges, ...) always lives in LDS (SmallWs /
// the KAT kernels' __shared__ copies).  The engine passes it around as generic pointers, for which the compiler emits flat_load/flat_store with
// 64-bit address arithmetic and a full s_waitcnt per access; the hot loops below therefore re-type the
// pointer as an LDS (address space 3) pointer: ds_read/ds_write with 32-bit addressing and immediate
// offsets, several loads in flight.  On the host simulation the qualifier is empty.
#if TK_HOST
#define TK_LDS
#else
#define TK_LDS __attribute__((address_space(3)))
#endif
typedef TK_LDS int16_t lds_i16;
#define TK_LDS_PTR(p) ((lds_i16*)(p))
// Same for data that always lives in global memory (frame planes, the per-wave BigWs sample blocks): pointers that come
// out of FrameJob / TeamWs are generic to the compiler, which then emits flat_load/flat_store (they tick the LDS counter
// as well as the vector-memory counter, so every LDS wait also waits for them); re-typed as address-space-1 pointers the
// hot loops use global_load/global_store.
#if TK_HOST
#define TK_GLOBAL
#else
#define TK_GLOBAL __attribute__((address_space(1)))
#endif
// Address space of a block of data as a compile-time property.  The engine's pointers are generic; a generic load/store
// (flat_*) of data that lives in LDS goes through the texture path (several hundred cycles, ticks vmcnt AND lgkmcnt) where a
// ds_read takes ~100.  The sample blocks / original samples / chroma coefficient buffers of a coding block live in LDS for
// blocks up to kLdsBlk and in global memory above; the decision code is therefore instantiated twice (SP = SP_LDS /
// SP_GLOBAL, chosen once per block decision) and the leaf loops re-type their pointers with spc<SP>().  Workspace structures
// that ALWAYS live in LDS on the device (XformWs, MeWs, WgShared, ...) are re-typed with ldsc().  Identity on the host.
enum { SP_GLOBAL = 0, SP_LDS = 1 };
// Wavefronts per SIMD the superblock kernel's register allocation is sized for (thor_hip.cpp:k_superblocks): 3 (168 VGPRs, three
// workgroups per CU, 53 KB of LDS each) or 2 (256 VGPRs, two workgroups per CU, 80 KB of LDS each).
#ifndef TK_OCC
#define TK_OCC 3
#endif
#if TK_HOST
template <int SP, class T> TK_DEV T* spc(T* p) { return p; }
template <class T> TK_DEV T* ldsc(T* p) { return p; }
TK_DEV int tk_is_lds(const void*) { return 0; }
#else
template <int SP, class T> struct SpT;
template <class T> struct SpT<0, T> { typedef TK_GLOBAL T* ptr; };
template <class T> struct SpT<1, T> { typedef TK_LDS T* ptr; };
template <int SP, class T> TK_DEV typename SpT<SP, T>::ptr spc(T* p) {
  if constexpr (SP == SP_LDS) return (TK_LDS T*)(uint32_t)(uintptr_t)p;  // generic LDS address = aperture | offset
  else return (TK_GLOBAL T*)p;
}
template <class T> TK_DEV TK_LDS T* ldsc(T* p) { return (TK_LDS T*)(uint32_t)(uintptr_t)p; }
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
TK_DEV int tk_is_lds(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void*)p) ? 1 : 0;
#else
  (void)p;
  return 0;  // host pass of the HIP compilation: never executed
#endif
}
#endif
// Whole-object load / store of a trivially copyable object that lives in LDS on the device (ds_read / ds_write of its bytes).
template <class T> TK_DEV T lds_ld(const T* p) {
  T v;
  __builtin_memcpy(&v, ldsc(p), sizeof(T));
  return v;
}
template <class T> TK_DEV void lds_st(T* p, const T& v) { __builtin_memcpy(ldsc(p), &v, sizeof(T)); }
#if !TK_HOST
template <class T> TK_DEV T lds_ld(const TK_LDS T* p) {  // already LDS-typed
  T v;
  __builtin_memcpy(&v, p, sizeof(T));
  return v;
}
template <class T> TK_DEV void lds_st(TK_LDS T* p, const T& v) { __builtin_memcpy(p, &v, sizeof(T)); }
#endif
template <class T> TK_DEV const TK_GLOBAL T* gptr(const T* p) { return (const TK_GLOBAL T*)p; }
template <class T> TK_DEV TK_GLOBAL T* gptr(T* p) { return (TK_GLOBAL T*)p; }
typedef uint32_t __attribute__((aligned(1), may_alias)) u32_unaligned;
typedef unsigned long long __attribute__((aligned(1), may_alias)) u64_unaligned;
TK_DEV uint32_t gload32(const void* p) { return *(const TK_GLOBAL u32_unaligned*)p; }
TK_DEV unsigned long long gload64(const void* p) { return *(const TK_GLOBAL u64_unaligned*)p; }

// scan index -> coefficient position for a qsize x qsize block (qsize 4, 8 or 16)
struct IzzRef {
#if TK_HOST
  const int16_t* z;
#else
  const lds_i16* z;
#endif
};
template <typename T> TK_DEV T tmin(T a, T b) { return a < b ? a : b; }
template <typename T> TK_DEV T tmax(T a, T b) { return a > b ? a : b; }
TK_DEV int iabs(int a) { return a < 0 ? -a : a; }
TK_DEV int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
TK_DEV int ilog2(unsigned x) {
#if TK_HOST
  return 31 - __builtin_clz(x);
#else
  return 31 - __clz((int)x);
#endif
}
// k -> (k / w, k % w) without an integer division when w is a power of two (the common case: every
// block / PU / TU dimension; only the frame-edge rectangular blocks are not).
struct Div2 {
  int w, sh;
};
TK_DEV Div2 mk_div(int w) {
  Div2 d;
  d.w = w;
  d.sh = (w & (w - 1)) ? -1 : ilog2((unsigned)w);
  return d;
}
TK_DEV void split2(const Div2& d, int k, int& i, int& j) {
  if (d.sh >= 0) { i = k >> d.sh; j = k & (d.w - 1); }
  else { i = k / d.w; j = k - i * d.w; }
}
// Same for dimensions that are powers of two by construction (TU / PU / CB sizes): no division fallback is
// compiled into the loop.
struct Pow2 {
  int sh, mask;
};
TK_DEV Pow2 mk_pow2(int w) {
  Pow2 d;
  d.sh = ilog2((unsigned)w);
  d.mask = w - 1;
  return d;
}
TK_DEV void split2(const Pow2& d, int k, int& i, int& j) {
  i = k >> d.sh;
  j = k & d.mask;
}
TK_DEV int sat_pix(int v, int bitdepth) { return clampi(v, 0, (1 << bitdepth) - 1); }

// ---------------------------------------------------------------------------------
// Codec constants (reference: common/global.h:54-93).
// ---------------------------------------------------------------------------------
enum { kMaxSb = 128, kMinBlk = 8, kMinPb = 4, kMaxQuant = 16, kPadY = 160, kMaxRefs = 4 };
enum { F_I = 0, F_P = 1, F_B = 2 };
enum { M_SKIP = 0, M_INTRA = 1, M_INTER = 2, M_BIPRED = 3, M_MERGE = 4 };
enum { P_NONE = 0, P_HOR = 1, P_VER = 2, P_QUAD = 3 };
enum { kNumIntraModes = 10 };
enum { kCostInit = 1u << 31 };  // reference MAX_UINT32 is 1<<31 (common/global.h:63)

struct mv_t {
  int16_t x, y;
};
TK_DEV mv_t mk_mv(int x, int y) {
  mv_t m;
  m.x = (int16_t)x;
  m.y = (int16_t)y;
  return m;
}

// Candidate / neighbour motion record (reference inter_pred_t, common/types.h:138-145).
// dir: 0 uni, 2 bi, -1 "came from an intra block" (reference stores (uint32_t)-1).
struct InterPred {
  mv_t mv0, mv1;
  int8_t ref0, ref1, dir, pad;
};

// Per 4x4-pel cell state (compact device form of deblock_data_t, common/types.h:178-187).
struct DbCell {
  mv_t mv0, mv1;
  uint8_t mode;
  uint8_t size;
  uint8_t tbpb;  // bit0 tb_split, bits1-2 pb_part
  uint8_t cbp;   // bit0 y>0, bit1 u>0, bit2 v>0
  int8_t ref0, ref1, dir;
  uint8_t pad;
};

// Constant tables: filled once by the host (tk_tables.h:init_tables) and uploaded
// (device) or used in place (hostsim).
struct Tables {
  int16_t zz4[16], zz8[64], zz16[256];       // position -> scan index (common_tables.c:29-66)
  int16_t izz4[16], izz8[64], izz16[256];    // scan index -> position
  int16_t dct4[16], dct8[64], dct16[256], dct32[1024];  // HEVC integer DCT (transform.c:37-241)
  uint8_t chroma_qp[52];
  uint8_t beta[52];
  uint8_t tc[56];
  uint16_t iq_8x8[52];                       // top-down split threshold scale (encode_block.c:2394-2398)
};

#if TK_HOST
extern Tables g_tab;
#define TK_TAB (tk::g_tab)
#else
extern __device__ Tables g_tab;
#define TK_TAB (tk::g_tab)
#endif

TK_DEV IzzRef izz_ref(const Team t, int qsize) {
  IzzRef r;
#if TK_HOST
  (void)t;
  r.z = qsize == 4 ? TK_TAB.izz4 : (qsize == 8 ? TK_TAB.izz8 : TK_TAB.izz16);
#else
  r.z = TK_LDS_PTR(t.izz) + (qsize == 4 ? 0 : (qsize == 8 ? 16 : 80));
#endif
  return r;
}


// quant / dequant scales (common_tables.c:74-75)
TK_DEV int quant_scale(int r) {
  return r == 0 ? 26214 : r == 1 ? 23302 : r == 2 ? 20560 : r == 3 ? 18396 : r == 4 ? 16384 : 14564;
}
TK_DEV int dequant_scale(int r) { return r == 0 ? 40 : r == 1 ? 45 : r == 2 ? 51 : r == 3 ? 57 : r == 4 ? 64 : 72; }

// ---------------------------------------------------------------------------------
// Frame / stream description shared by host and device.
// ---------------------------------------------------------------------------------
template <typename PIX> struct Plane3 {
  PIX* y;
  PIX* u;
  PIX* v;    // pointers to pixel (0,0) (inside the padding for padded planes)
  int sy, sc;  // strides in samples
};

struct EncCfg {  // the subset of enc_params the block path reads (enc/mainenc.h:35-112)
  int width, height;
  int bitdepth;
  int enable_tb_split, enable_pb_split, enable_bipred;
  int encoder_speed, intra_rdo, use_block_contexts;
  int cfl_intra, cfl_inter;
  int max_num_ref, interp_ref_cfg;
  float early_skip_thr;
};

template <typename PIX> struct FrameJob {
  EncCfg cfg;
  int frame_type, qp, num_ref, frame_num, interp_ref, num_intra_modes;
  int sign[kMaxRefs];     // ref->frame_num >  cur frame_num  (uni-pred / RDO paths)
  int sign_ge[kMaxRefs];  // ref->frame_num >= cur frame_num  (bi-pred early skip, Appendix B.19)
  double lambda;          // lambda_coeff * squared_lambda_QP[qp]
  double sqrt_lambda;     // sqrt(lambda), computed on the host with libm
  Plane3<PIX> orig, rec;
  Plane3<PIX> ref[kMaxRefs];  // by ref_idx (already resolved through ref_array[])
  DbCell* cells;              // (height/4) x (width/4)
  int cell_stride;
  int sb_cols, sb_rows;
  uint32_t* sb_bits;       // per SB bit buffer, sb_words words each
  int sb_words;
  int* sb_nbits;           // per SB number of bits written
  int* sb_status;          // 0 ok, !=0 overflow/internal error
  uint8_t* scratch;        // per-team scratch arena, scratch_bytes each
  size_t scratch_bytes;
  long long* prof;         // optional cycle-counter sink (THOR_PROF builds), 16 slots
  // content statistics of the inter frames (SURVEY 8d "always log the fraction of SBs that early-skipped"), engine-wide,
  // may be nullptr: [0] luma pixels coded by the early-skip shortcut (any block size), [1] superblocks that early-skipped
  // as a whole 128x128 block, [2] superblocks processed, [3] luma pixels processed
  unsigned long long* stats;
};

}  // namespace tk
