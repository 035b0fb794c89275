// thor_hip.cpp - libthor_hip.so: gfx950 kernels, device backend and the C ABI (include/thor_hip.h).
// One workgroup of 4 wavefronts encodes one 128x128 superblock at a time (wave 0 walks the quadtree, all four share the
// trials of each block decision); one persistent, dependency-driven launch per frame covers every superblock of every stream (SB(k,l) needs (k,l-1) and
// (k-1,l+1)).  There is NO CPU path in this library: every entry point aborts if no
// HIP device is usable.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <vector>

#include "tk_block.h"
#include "tk_filters.h"
#include "tk_cdef.h"
#include "tk_clpf.h"
#include "tk_encoder.h"
#include "tk_cli.h"
#include "tk_sched.h"
#include "tk_kernel.h"
#include "../../include/thor_hip.h"

#define HIPCHECK(x)                                                                              \
  do {                                                                                           \
    hipError_t e_ = (x);                                                                         \
    if (e_ != hipSuccess) {                                                                      \
      fprintf(stderr, "Run-time error...\nthor_hip: %s failed: %s (%s:%d)\n...now exiting to system...\n", #x, \
              hipGetErrorString(e_), __FILE__, __LINE__);                                        \
      abort();                                                                                   \
    }                                                                                            \
  } while (0)

// the second build of the engine (thor_hip_lat.cpp: 256 VGPRs, two workgroups per CU - the few-stream operating point)
#define TK_ALT_DECLS(P)                                                                                                                  \
  extern "C" __attribute__((visibility("hidden"))) int P##upload_tables(const void* tables, size_t bytes);                                 \
  extern "C" __attribute__((visibility("hidden"))) int P##workgroups_per_cu(void);                                                         \
  extern "C" __attribute__((visibility("hidden"))) int P##waves(void);                                                                     \
  extern "C" __attribute__((visibility("hidden"))) int P##kernel_info(int* num_regs, int* lds_bytes, int* private_bytes);                  \
  extern "C" __attribute__((visibility("hidden"))) int P##launch_u8(int wgs, void* stream, const void* jobs, const void* dfargs, size_t dfargs_bytes, size_t job_bytes, size_t slot_bytes);
TK_ALT_DECLS(thor_lat_)    // thor_hip_lat.cpp: 256 VGPRs, two four-wave workgroups per CU
TK_ALT_DECLS(thor_wide_)   // thor_hip_wide.cpp: eight-wave workgroups, one per CU

namespace tk {
__device__ Tables g_tab;

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// Dependency-driven persistent superblock kernel.  A task is (stream, superblock).  SB(k,l) needs its left
// neighbour (k,l-1) and its up-right neighbour (k-1,l+1) ((k-1,l) in the last column) - SURVEY.md Appendix A.
// The ready-task queue is in tk_sched.h.

template <typename PIX> __global__ void k_deblock(const FrameJob<PIX>* jobs, int pass) {
  const FrameJob<PIX>& J = jobs[blockIdx.y];
  DbParams P;
  P.width = J.cfg.width; P.height = J.cfg.height; P.bitdepth = J.cfg.bitdepth;
  const int qpc = g_tab.chroma_qp[J.qp];
  P.beta = g_tab.beta[J.qp] << (P.bitdepth - 8);
  P.tc_y = g_tab.tc[J.qp] >> (12 - P.bitdepth);
  P.tc_c = g_tab.tc[qpc] >> (12 - P.bitdepth);
  P.cells = J.cells; P.cs = J.cell_stride;
  deblock_pass(J.rec, P, pass, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x));
}

template <typename PIX> struct RefJob { Plane3<PIX> rec, ref; int width, height; };
template <typename PIX> __global__ void k_make_ref(const RefJob<PIX>* rj) {
  const RefJob<PIX>& R = rj[blockIdx.y];
  make_ref_rows(R.rec, R.ref, R.width, R.height, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x, (int)blockDim.x);
}

// Bit-level concatenation: one workgroup per item.  dst is zero-filled; words are OR-ed in.
__global__ void k_gather_bits(const backend::GatherItem* items, int n, uint32_t* dst) {
  const int it = blockIdx.x;
  if (it >= n) return;
  const backend::GatherItem g = items[it];
  const int nw = (g.nbits + 31) >> 5;
  const int sh = (int)(g.dst_bit & 31);
  const long long w0 = g.dst_bit >> 5;
  for (int j = threadIdx.x; j < nw; j += blockDim.x) {
    uint32_t v = g.src[j];
    const int valid = g.nbits - 32 * j;           // bits of this word that belong to the string
    if (valid < 32) v &= ~((1u << (32 - valid)) - 1u);
    if (sh == 0) atomicOr(&dst[w0 + j], v);
    else {
      atomicOr(&dst[w0 + j], v >> sh);
      const uint32_t lo = v << (32 - sh);
      if (lo) atomicOr(&dst[w0 + j + 1], lo);
    }
  }
}

template <typename PIX> __global__ void k_copy_planes(const CdefJob<PIX>* cj) {
  const CdefJob<PIX>& C = cj[blockIdx.y];
  const int rows = C.height + C.height;  // Y rows + U rows + V rows
  for (int it = blockIdx.x; it < rows; it += gridDim.x) {
    const PIX* s; PIX* d; int w;
    if (it < C.height) { s = C.rec.y + (size_t)it * C.rec.sy; d = C.src.y + (size_t)it * C.src.sy; w = C.width; }
    else if (it < C.height + C.height / 2) { int r = it - C.height; s = C.rec.u + (size_t)r * C.rec.sc; d = C.src.u + (size_t)r * C.src.sc; w = C.width / 2; }
    else { int r = it - C.height - C.height / 2; s = C.rec.v + (size_t)r * C.rec.sc; d = C.src.v + (size_t)r * C.src.sc; w = C.width / 2; }
    for (int x = threadIdx.x; x < w; x += blockDim.x) d[x] = s[x];
  }
}
// passes 0 (flags), 1 (direction / variance per 8x8 block) and 4 (apply) of CDEF: one instance per pass, so that each gets its own register allocation
template <typename PIX, int PASS> __global__ __launch_bounds__(256) void k_cdef(const CdefJob<PIX>* cj) {   // (no bound = 1024 threads = a 128-VGPR cap: the apply pass spilled 30)
  const CdefJob<PIX>& C = cj[blockIdx.y];
  const int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x), gsize = (int)(gridDim.x * blockDim.x);
  if constexpr (PASS == 0) cdef_pass_flags(C, gid, gsize);
  else if constexpr (PASS == 1) cdef_pass_dir(C, gid, gsize);
  else cdef_pass_apply(C, gid, gsize);
}
// pass 2 of the CDEF search (tk_cdef.h: wavefront form): one wavefront per 8x8 luma-unit block, four independent wavefronts per workgroup (no workgroup
// barrier: a wavefront whose block is skipped leaves at once)
template <typename PIX> __global__ __launch_bounds__(256) void k_cdef_mse(const CdefJob<PIX>* cj) {
  const CdefJob<PIX>& C = cj[blockIdx.y];
  if (!C.cdef_bits) return;
  __shared__ CdefWaveWs<PIX> ws[4];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
  const int b = (int)blockIdx.x * 4 + wave;
  if (b >= (C.width / 8) * (C.height / 8)) return;
  cdef_mse_block_wave(mk_team(lane, 64), C, b, &ws[wave]);
}
template <typename PIX> __global__ void k_clpf(const ClpfJob<PIX>* lj, int pass) {
  const ClpfJob<PIX>& L = lj[blockIdx.y];
  const int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x), gsize = (int)(gridDim.x * blockDim.x);
  if (pass == 0) clpf_pass_stats(L, gid, gsize);
  else clpf_pass_apply(L, gid, gsize);
}
template <typename PIX> __global__ void k_clpf_copy(const ClpfJob<PIX>* lj) {  // rec -> src (unfiltered copy)
  const ClpfJob<PIX>& C = lj[blockIdx.y];
  const int rows = C.height + C.height;
  for (int it = blockIdx.x; it < rows; it += gridDim.x) {
    const PIX* s; PIX* d; int w;
    if (it < C.height) { s = C.rec.y + (size_t)it * C.rec.sy; d = C.src.y + (size_t)it * C.src.sy; w = C.width; }
    else if (it < C.height + C.height / 2) { int r = it - C.height; s = C.rec.u + (size_t)r * C.rec.sc; d = C.src.u + (size_t)r * C.src.sc; w = C.width / 2; }
    else { int r = it - C.height - C.height / 2; s = C.rec.v + (size_t)r * C.rec.sc; d = C.src.v + (size_t)r * C.src.sc; w = C.width / 2; }
    for (int x = threadIdx.x; x < w; x += blockDim.x) d[x] = s[x];
  }
}
template <typename PIX> __global__ __launch_bounds__(1024) void k_cdef_select(const CdefJob<PIX>* cj) {
  BlockTeam t{(int)threadIdx.x, (int)blockDim.x};
  cdef_pass_select(t, cj[blockIdx.x]);
}

// ---- temporally interpolated reference (tk_interp_dev.h) -----------------------------------------------------------
template <typename PIX> __global__ void k_interp_clear(const idev::Job<PIX>* jobs) {
  const idev::Job<PIX>& J = jobs[blockIdx.y];
  const int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x), gsz = (int)(gridDim.x * blockDim.x);
  for (int l = 0; l < J.levels; l++) {
    const idev::Level<PIX>& L = J.lv[l];
    const int cnt = L.bw * L.bh + L.bw + 2;
    uint32_t* a = (uint32_t*)L.mv[0];
    uint32_t* b = (uint32_t*)L.mv[1];
    for (int k = gid; k < cnt; k += gsz) { a[k] = 0; b[k] = 0; }
    for (int k = gid; k < L.bh / idev::kStep + 1; k += gsz) L.prog[k] = 0;
  }
}
template <typename PIX> __global__ void k_interp_down(const idev::Job<PIX>* jobs, int l) {
  const idev::Job<PIX>& J = jobs[blockIdx.y];
  if (l >= J.levels) return;
  const int ow = J.width >> l, oh = J.height >> l, pw = ow + 64;
  const int total = (oh + 64) * pw;
  for (int k = (int)(blockIdx.x * blockDim.x + threadIdx.x); k < total; k += (int)(gridDim.x * blockDim.x)) {
    const int i = k / pw - 32, j = k % pw - 32;
    for (int r = 0; r < 2; r++)
      idev::down2x2_item(l == 1 ? J.ref[r].y : J.dpic[r][l - 1], l == 1 ? J.ref[r].sy : J.dstride[l - 1], J.dpic[r][l], J.dstride[l], ow, oh, i, j);
  }
}
// One wavefront per 16x16-block row.  Rows are handed out by a ticket, so the row above a wave's row was always taken by
// a wave that started earlier: the wave waits until that row is two blocks ahead (or finished) and never dead-locks.
template <typename PIX> __global__ __launch_bounds__(64) void k_interp_estimate(const idev::Job<PIX>* jobs, int lvl) {
  const idev::Job<PIX>& J = jobs[blockIdx.y];
  if (lvl >= J.levels) return;
  const idev::Level<PIX>& L = J.lv[lvl];
  const Team t = mk_team((int)threadIdx.x, 64);
  int row = 0;
  if (threadIdx.x == 0) row = (int)atomicAdd((unsigned*)L.ticket, 1u);
  row = __builtin_amdgcn_readfirstlane(row);
  const int nrows = L.bh / idev::kStep, ncols = L.bw / idev::kStep;
  if (row >= nrows) return;
  for (int c = 0; c < ncols; c++) {
    if (row > 0) {
      const int need = c + 2 < ncols ? c + 2 : ncols;
      while (__hip_atomic_load(&L.prog[row - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(8);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    idev::estimate_block(t, L, row * idev::kStep, c * idev::kStep);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (threadIdx.x == 0) __hip_atomic_store(&L.prog[row], c + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
template <typename PIX> __global__ __launch_bounds__(64) void k_interp_merge(const idev::Job<PIX>* jobs, int lvl) {
  const idev::Job<PIX>& J = jobs[blockIdx.y];
  if (lvl >= J.levels) return;
  const idev::Level<PIX>& L = J.lv[lvl];
  const Team t = mk_team((int)threadIdx.x, 64);
  for (int k = blockIdx.x; k < L.bw * L.bh; k += gridDim.x) idev::merge_block(t, L, k / L.bw, k % L.bw);
}
template <typename PIX> __global__ void k_interp_upscale(const idev::Job<PIX>* jobs, int lvl) {  // level lvl -> guide of lvl-1
  const idev::Job<PIX>& J = jobs[blockIdx.y];
  if (lvl >= J.levels || lvl < 1) return;
  const idev::Level<PIX>& L = J.lv[lvl];
  const idev::Level<PIX>& O = J.lv[lvl - 1];
  for (int k = (int)(blockIdx.x * blockDim.x + threadIdx.x); k < O.bw * O.bh; k += (int)(gridDim.x * blockDim.x))
    idev::upscale_item(L.nmv[1], L.bw, O.gmv1, O.bw, k / O.bw, k % O.bw);
}
template <typename PIX> __global__ __launch_bounds__(64) void k_interp_mc(const idev::Job<PIX>* jobs) {
  const idev::Job<PIX>& J = jobs[blockIdx.y];
  const idev::Level<PIX>& L = J.lv[0];
  const Team t = mk_team((int)threadIdx.x, 64);
  for (int k = blockIdx.x; k < L.bw * L.bh; k += gridDim.x) idev::mot_comp_unit(t, J, k / L.bw, k % L.bw);
}
template <typename PIX> __global__ void k_interp_pad(const idev::Job<PIX>* jobs) {
  const idev::Job<PIX>& J = jobs[blockIdx.y];
  idev::pad_item(J, (int)blockIdx.x, (int)threadIdx.x, (int)blockDim.x);
}

// ---------------------------------------------------------------------------------------------
// backend
// ---------------------------------------------------------------------------------------------
static bool g_inited = false;
static hipStream_t g_stream = nullptr;
struct KernelClock {
  std::vector<hipEvent_t> ev;  // pairs
  double sb_ms = 0, filt_ms = 0;
  long sb_launches = 0;
};
static KernelClock g_clk;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_sb_events, g_filt_events;

// The backend is one-device-per-process and single-threaded by design (one process per GPU, include/thor_hip.h): the
// first call fixes the device; a later request for another device, or a device index the node does not have, is an
// error (returns false) - never a silent fall-back to device 0.
static int g_device = -1;
static bool ensure_init(int device) {
  if (g_inited) {
    if (device != g_device) { fprintf(stderr, "thor_hip: this process is bound to HIP device %d (requested %d); use one process per GPU\n", g_device, device); return false; }
    HIPCHECK(hipSetDevice(g_device));  // hipSetDevice is per thread
    return true;
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    fprintf(stderr, "Run-time error...\nthor_hip: no HIP device available - this library has no CPU path\n...now exiting to system...\n");
    abort();
  }
  if (device < 0 || device >= n) { fprintf(stderr, "thor_hip: HIP device %d requested but only %d visible (check LOCAL_RANK / HIP_VISIBLE_DEVICES)\n", device, n); return false; }
  g_device = device;
  HIPCHECK(hipSetDevice(device));
  HIPCHECK(hipStreamCreate(&g_stream));
  static Tables h;
  init_tables(&h);
  HIPCHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_tab), &h, sizeof(h)));
  if (thor_lat_upload_tables(&h, sizeof(h)) || thor_wide_upload_tables(&h, sizeof(h))) { fprintf(stderr, "Run-time error...\nthor_hip: table upload of the few-stream kernels failed\n...now exiting to system...\n"); abort(); }
  g_inited = true;
  return true;
}

namespace backend {
void* dev_alloc(size_t n) {
  void* p = nullptr;
  HIPCHECK(hipMalloc(&p, n ? n : 1));
  HIPCHECK(hipMemsetAsync(p, 0, n ? n : 1, g_stream));
  return p;
}
void dev_free(void* p) { if (p) HIPCHECK(hipFree(p)); }
void h2d(void* d, const void* h, size_t n) { HIPCHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, g_stream)); HIPCHECK(hipStreamSynchronize(g_stream)); }
void d2h(void* h, const void* d, size_t n) { HIPCHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, g_stream)); HIPCHECK(hipStreamSynchronize(g_stream)); }
void dev_memset(void* d, int v, size_t n) { HIPCHECK(hipMemsetAsync(d, v, n, g_stream)); }
static void harvest(std::vector<std::pair<hipEvent_t, hipEvent_t>>& v, double& acc) {
  for (auto& p : v) {
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, p.first, p.second));
    acc += ms;
    HIPCHECK(hipEventDestroy(p.first));
    HIPCHECK(hipEventDestroy(p.second));
  }
  v.clear();
}
void dev_sync() {
  HIPCHECK(hipStreamSynchronize(g_stream));
  harvest(g_sb_events, g_clk.sb_ms);
  harvest(g_filt_events, g_clk.filt_ms);
}
size_t team_ws_bytes(int pix_bytes) { return pix_bytes == 1 ? sizeof(BigWs<uint8_t>) : sizeof(BigWs<uint16_t>); }

static std::pair<hipEvent_t, hipEvent_t> ev_begin() {
  std::pair<hipEvent_t, hipEvent_t> p;
  HIPCHECK(hipEventCreate(&p.first));
  HIPCHECK(hipEventCreate(&p.second));
  HIPCHECK(hipEventRecord(p.first, g_stream));
  return p;
}

struct DfState {  // per engine (keyed by its device job array)
  DfCtl* ctl = nullptr;
  unsigned* queue = nullptr;
  unsigned* cnt = nullptr;
  uint8_t* pool = nullptr;
  unsigned long long* times = nullptr;
  unsigned* range = nullptr;
  int S = 0, nsb = 0, wgs = 0;
  size_t slot = 0;
  int frame = 0;
  int kern = 0;  // the build of the 8-bit kernel this engine's launches use: 0 thor_hip.cpp (throughput), 1 thor_hip_lat.cpp, 2 thor_hip_wide.cpp
};
static std::map<const void*, DfState> g_df;
static int g_last_kern = 0;   // build of the 8-bit kernel the most recently configured engine uses (thor_hip_superblock_kernel_in_use)
static void df_free(DfState& D) {
  if (!D.ctl) return;
  HIPCHECK(hipFree(D.ctl)); HIPCHECK(hipFree(D.queue)); HIPCHECK(hipFree(D.cnt)); HIPCHECK(hipFree(D.pool)); HIPCHECK(hipFree(D.range));
  if (D.times) HIPCHECK(hipFree(D.times));
}
template <typename PIX> void run_superblocks(const FrameJob<PIX>* jobs, const FrameJob<PIX>* hjobs, int S, const SbRange* ranges) {
  const int cols = hjobs[0].sb_cols, rows = hjobs[0].sb_rows, nsb = cols * rows;
  const size_t all = (size_t)S * nsb;
  DfState& D = g_df[jobs];
  const size_t slot = (sizeof(BigWs<PIX>) + 255) & ~(size_t)255;
  if (D.S != S || D.nsb != nsb || D.slot != slot) {
    df_free(D);
    D = DfState();
    D.S = S; D.nsb = nsb; D.slot = slot;
    int per_cu = 0;
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_superblocks<PIX>, kWgThreads, 0));
    hipDeviceProp_t prop;
    int dev = 0;
    HIPCHECK(hipGetDevice(&dev));
    HIPCHECK(hipGetDeviceProperties(&prop, dev));
    long cap = (long)(per_cu > 0 ? per_cu : 1) * prop.multiProcessorCount;
    // Which build of the kernel: a stream offers at most min(rows, (cols + 1) / 2) superblocks at a time (the dependency wavefront).  Eight-wave workgroups
    // (one per CU) finish a superblock ~25 % sooner than four-wave ones and saturate at ~3/4 of the throughput build's peak, which that build only reaches
    // with well over a hundred streams: measured (round 6, call 13) the wide build wins by 24-26 % up to S x wavefront = 2 x CUs (3840x2160: 24 / 32 streams
    // 61.3 / 80.2 against 49.2 / 64.1 Mpx/s with four-wave workgroups, 1920x1080: 32 / 64 streams 40.8 / 76.9 against 32.9 / 61.6) and still by 5 % at 2.8 x CUs
    // (48 streams at 3840x2160: 96.3 against 91.5).  Rule: wide up to 2.5 x CUs, the throughput build above.  The latency build (256 VGPRs, two four-wave
    // workgroups per CU) lost its range to the wide build and runs only when forced.  8-bit samples only (the 16-bit kernel has one build).
    // THOR_HIP_KERNEL=std|lat|wide forces one (tests, A/B).
    int pool_waves = kWaves;
    if constexpr (sizeof(PIX) == 1) {
      const long lat_cap = (long)thor_lat_workgroups_per_cu() * prop.multiProcessorCount;
      const long wide_cap = (long)thor_wide_workgroups_per_cu() * prop.multiProcessorCount;
      const long wave_front = (long)S * (rows < (cols + 1) / 2 ? rows : (cols + 1) / 2);
      D.kern = wide_cap > 0 && 2 * wave_front <= 5 * wide_cap ? 2 : 0;
      if (const char* e = getenv("THOR_HIP_KERNEL")) {
        if (!strcmp(e, "lat")) D.kern = lat_cap > 0 ? 1 : 0;
        else if (!strcmp(e, "wide")) D.kern = wide_cap > 0 ? 2 : 0;
        else if (!strcmp(e, "std")) D.kern = 0;
      }
      if (D.kern == 1) cap = lat_cap;
      if (D.kern == 2) { cap = wide_cap; pool_waves = thor_wide_waves(); }
    }
    g_last_kern = sizeof(PIX) == 1 ? D.kern : 0;
    if (const char* e = getenv("THOR_HIP_WGS")) { if (*e) cap = atol(e); }
    D.wgs = (int)(cap < (long)all ? cap : (long)all);
    HIPCHECK(hipMalloc(&D.ctl, sizeof(DfCtl)));
    HIPCHECK(hipMalloc(&D.queue, sizeof(unsigned) * all));
    HIPCHECK(hipMalloc(&D.cnt, sizeof(unsigned) * all));
    HIPCHECK(hipMalloc(&D.range, sizeof(unsigned) * S));
    HIPCHECK(hipMalloc(&D.pool, slot * (size_t)D.wgs * pool_waves));  // one BigWs slot per wavefront
    if (getenv("THOR_SBTIMES")) { HIPCHECK(hipMalloc(&D.times, sizeof(unsigned long long) * 3 * all)); }
  }
  // launch start: the superblocks of every stream's range whose dependencies all lie below the range (whole frames: SB(0,0))
  size_t total = 0;
  {
    std::vector<unsigned> q0, hr(S);
    for (int s2 = 0; s2 < S; s2++) {
      const int lo = ranges ? ranges[s2].lo : 0, hi = ranges ? ranges[s2].hi : 0x7fff;
      hr[s2] = (unsigned)lo | ((unsigned)hi << 16);
      if (lo >= hi) continue;
      for (int k = 0; k < rows; k++)
        for (int l = 0; l < cols; l++) {
          const int t = df_diag(k, l);
          if (t < lo || t >= hi) continue;
          total++;
          if (df_need(k, l, cols, lo) == 0) q0.push_back((unsigned)s2 * (unsigned)nsb + (unsigned)(k * cols + l));
        }
    }
    if (!total) return;
    DfCtl hc0 = {0u, (unsigned)q0.size(), 0u, 0u};
    HIPCHECK(hipMemsetAsync(D.queue, 0xff, sizeof(unsigned) * total, g_stream));
    HIPCHECK(hipMemsetAsync(D.cnt, 0, sizeof(unsigned) * all, g_stream));
    if (D.times) HIPCHECK(hipMemsetAsync(D.times, 0, sizeof(unsigned long long) * 3 * all, g_stream));
    HIPCHECK(hipMemcpyAsync(D.queue, q0.data(), sizeof(unsigned) * q0.size(), hipMemcpyHostToDevice, g_stream));
    HIPCHECK(hipMemcpyAsync(D.range, hr.data(), sizeof(unsigned) * S, hipMemcpyHostToDevice, g_stream));
    HIPCHECK(hipMemcpyAsync(D.ctl, &hc0, sizeof(hc0), hipMemcpyHostToDevice, g_stream));
    HIPCHECK(hipStreamSynchronize(g_stream));
  }
  DfArgs A;
  A.ctl = D.ctl; A.queue = D.queue; A.cnt = D.cnt; A.pool = D.pool; A.slot_bytes = slot; A.times = D.times;
  A.S = S; A.nsb = nsb; A.cols = cols; A.rows = rows;
  A.range = ranges ? D.range : nullptr; A.total = (unsigned)total;
  double lim_s = 300.0;
  if (const char* e = getenv("THOR_HIP_SPIN_TIMEOUT_S")) lim_s = atof(e);
  A.spin_limit = (unsigned long long)(lim_s * 1e8);
  auto ev = ev_begin();
  if (D.kern) {
    if ((D.kern == 2 ? thor_wide_launch_u8 : thor_lat_launch_u8)(D.wgs, (void*)g_stream, jobs, &A, sizeof(A), sizeof(FrameJob<PIX>), slot)) { fprintf(stderr, "Run-time error...\nthor_hip: launch of a few-stream kernel failed\n...now exiting to system...\n"); abort(); }
  } else
    hipLaunchKernelGGL(k_superblocks<PIX>, dim3(D.wgs), dim3(kWgThreads), 0, g_stream, jobs, A);
  g_clk.sb_launches++;
  HIPCHECK(hipEventRecord(ev.second, g_stream));
  g_sb_events.push_back(ev);
  HIPCHECK(hipGetLastError());
  DfCtl hc;
  HIPCHECK(hipMemcpyAsync(&hc, D.ctl, sizeof(hc), hipMemcpyDeviceToHost, g_stream));
  if (hipError_t e = hipStreamSynchronize(g_stream)) {
    // an aborted launch: the kernel traps when a wavefront waits for another wave of its workgroup beyond kWgWaitLimit
    // (tk_block.h:md_item_trial - a protocol error of the block decision's work queue, never a matter of load)
    fprintf(stderr, "Run-time error...\nthor_hip: k_superblocks was aborted: %s (a trap inside the kernel = an intra-workgroup wait that did not end)\n...now exiting to system...\n",
            hipGetErrorString(e));
    abort();
  }
  const unsigned handed_out = hc.tail;
  if (hc.error || handed_out != (unsigned)total) {
    fprintf(stderr, "Run-time error...\nthor_hip: superblock scheduler failed (error %u, %u of %zu tasks released)\n...now exiting to system...\n", hc.error, handed_out, total);
    abort();
  }
  if (D.times) {
    std::vector<unsigned long long> h(3 * all);   // superblocks outside this launch's ranges: zeros
    HIPCHECK(hipMemcpy(h.data(), D.times, h.size() * 8, hipMemcpyDeviceToHost));
    FILE* f = fopen(getenv("THOR_SBTIMES"), D.frame == 0 ? "wb" : "ab");
    if (f) { int hdr[4] = {D.frame, S, nsb, cols}; fwrite(hdr, 4, 4, f); fwrite(h.data(), 8, h.size(), f); fclose(f); }
  }
  D.frame++;
}
// Engine::close: the scheduler state belongs to the engine that owns `jobs`; without this a later engine whose job
// array lands on the same device address would inherit a pool sized for another sample type.
void release_superblocks(const void* jobs) {
  auto it = g_df.find(jobs);
  if (it == g_df.end()) return;
  DfState& D = it->second;
  df_free(D);
  g_df.erase(it);
}
template <typename PIX> void run_deblock(const FrameJob<PIX>* jobs, const FrameJob<PIX>* hjobs, int S) {
  const int items = (hjobs[0].cfg.width / 8) * (hjobs[0].cfg.height / 8);
  auto ev = ev_begin();
  for (int pass = 0; pass < 4; pass++)
    hipLaunchKernelGGL(k_deblock<PIX>, dim3((items + 63) / 64, S), dim3(64), 0, g_stream, jobs, pass);
  HIPCHECK(hipEventRecord(ev.second, g_stream));
  g_filt_events.push_back(ev);
  HIPCHECK(hipGetLastError());
}
template <typename PIX> void run_make_ref(const FrameJob<PIX>* hjobs, const Plane3<PIX>* dst, int S) {
  static RefJob<PIX>* d_rj = nullptr;
  static int cap = 0;
  std::vector<RefJob<PIX>> h(S);
  for (int s = 0; s < S; s++) { h[s].rec = hjobs[s].rec; h[s].ref = dst[s]; h[s].width = hjobs[s].cfg.width; h[s].height = hjobs[s].cfg.height; }
  if (cap < S) { if (d_rj) HIPCHECK(hipFree(d_rj)); HIPCHECK(hipMalloc(&d_rj, sizeof(RefJob<PIX>) * S)); cap = S; }
  HIPCHECK(hipMemcpyAsync(d_rj, h.data(), sizeof(RefJob<PIX>) * S, hipMemcpyHostToDevice, g_stream));
  HIPCHECK(hipStreamSynchronize(g_stream));  // h goes out of scope
  const int rows = hjobs[0].cfg.height * 2 + 4 * kPadY;
  auto ev = ev_begin();
  hipLaunchKernelGGL(k_make_ref<PIX>, dim3(rows, S), dim3(256), 0, g_stream, d_rj);
  HIPCHECK(hipEventRecord(ev.second, g_stream));
  g_filt_events.push_back(ev);
  HIPCHECK(hipGetLastError());
}
template <typename PIX> void run_cdef(const CdefJob<PIX>* cj, const CdefJob<PIX>* hcj, int S) {
  const int blocks8 = (hcj[0].width / 8) * (hcj[0].height / 8);
  auto ev = ev_begin();
  hipLaunchKernelGGL(k_copy_planes<PIX>, dim3(hcj[0].height * 2, S), dim3(256), 0, g_stream, cj);
  hipLaunchKernelGGL((k_cdef<PIX, 0>), dim3(64, S), dim3(256), 0, g_stream, cj);
  hipLaunchKernelGGL((k_cdef<PIX, 1>), dim3((blocks8 + 63) / 64, S), dim3(64), 0, g_stream, cj);
  hipLaunchKernelGGL(k_cdef_mse<PIX>, dim3((blocks8 + 3) / 4, S), dim3(256), 0, g_stream, cj);
  hipLaunchKernelGGL(k_cdef_select<PIX>, dim3(S), dim3(1024), 0, g_stream, cj);
  hipLaunchKernelGGL((k_cdef<PIX, 4>), dim3((blocks8 + 63) / 64, S), dim3(64), 0, g_stream, cj);
  HIPCHECK(hipEventRecord(ev.second, g_stream));
  g_filt_events.push_back(ev);
  HIPCHECK(hipGetLastError());
}
template <typename PIX> void run_clpf_stats(const ClpfJob<PIX>* lj, const ClpfJob<PIX>* hlj, int S) {
  const int nb = (hlj[0].width / 8) * (hlj[0].height / 8) + 2 * (hlj[0].width / 16) * (hlj[0].height / 16);
  auto ev = ev_begin();
  hipLaunchKernelGGL(k_clpf<PIX>, dim3((nb + 63) / 64, S), dim3(64), 0, g_stream, lj, 0);
  HIPCHECK(hipEventRecord(ev.second, g_stream));
  g_filt_events.push_back(ev);
  HIPCHECK(hipGetLastError());
}
template <typename PIX> void run_clpf_apply(const ClpfJob<PIX>* lj, const ClpfJob<PIX>* hlj, int S) {
  const int nu = 3 * (hlj[0].width / 8) * (hlj[0].height / 8);
  auto ev = ev_begin();
  hipLaunchKernelGGL(k_clpf_copy<PIX>, dim3(hlj[0].height * 2, S), dim3(256), 0, g_stream, lj);
  hipLaunchKernelGGL(k_clpf<PIX>, dim3((nu + 63) / 64, S), dim3(64), 0, g_stream, lj, 1);
  HIPCHECK(hipEventRecord(ev.second, g_stream));
  g_filt_events.push_back(ev);
  HIPCHECK(hipGetLastError());
}
template <typename PIX> void run_interp(const idev::Job<PIX>* jobs, const idev::Job<PIX>* hjobs, int n) {
  const idev::Job<PIX>& H = hjobs[0];  // all streams share the geometry
  auto ev = ev_begin();
  hipLaunchKernelGGL(k_interp_clear<PIX>, dim3(64, n), dim3(256), 0, g_stream, jobs);
  for (int l = 1; l < H.levels; l++) {
    const int total = ((H.height >> l) + 64) * ((H.width >> l) + 64);
    hipLaunchKernelGGL(k_interp_down<PIX>, dim3((total + 255) / 256, n), dim3(256), 0, g_stream, jobs, l);
  }
  for (int lvl = H.levels - 1; lvl >= 0; --lvl) {
    const idev::Level<PIX>& L = H.lv[lvl];
    hipLaunchKernelGGL(k_interp_estimate<PIX>, dim3(L.bh / idev::kStep, n), dim3(64), 0, g_stream, jobs, lvl);
    const int units = L.bw * L.bh;
    hipLaunchKernelGGL(k_interp_merge<PIX>, dim3(units < 16384 ? units : 16384, n), dim3(64), 0, g_stream, jobs, lvl);
    if (lvl > 0) {
      const int fine = H.lv[lvl - 1].bw * H.lv[lvl - 1].bh;
      hipLaunchKernelGGL(k_interp_upscale<PIX>, dim3((fine + 255) / 256, n), dim3(256), 0, g_stream, jobs, lvl);
    } else {
      hipLaunchKernelGGL(k_interp_mc<PIX>, dim3(units < 16384 ? units : 16384, n), dim3(64), 0, g_stream, jobs);
      const int rows = H.height + 2 * kPadY + 2 * (H.height / 2 + kPadY);
      hipLaunchKernelGGL(k_interp_pad<PIX>, dim3(rows, n), dim3(256), 0, g_stream, jobs);
    }
  }
  HIPCHECK(hipEventRecord(ev.second, g_stream));
  g_filt_events.push_back(ev);
  HIPCHECK(hipGetLastError());
}
template void run_interp<uint8_t>(const idev::Job<uint8_t>*, const idev::Job<uint8_t>*, int);
template void run_interp<uint16_t>(const idev::Job<uint16_t>*, const idev::Job<uint16_t>*, int);
void run_gather(const GatherItem* d_items, int n, uint32_t* dst) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather_bits, dim3(n), dim3(64), 0, g_stream, d_items, n, dst);
  HIPCHECK(hipGetLastError());
}
template void run_superblocks<uint8_t>(const FrameJob<uint8_t>*, const FrameJob<uint8_t>*, int, const SbRange*);
template void run_deblock<uint8_t>(const FrameJob<uint8_t>*, const FrameJob<uint8_t>*, int);
template void run_make_ref<uint8_t>(const FrameJob<uint8_t>*, const Plane3<uint8_t>*, int);
template void run_cdef<uint8_t>(const CdefJob<uint8_t>*, const CdefJob<uint8_t>*, int);
template void run_superblocks<uint16_t>(const FrameJob<uint16_t>*, const FrameJob<uint16_t>*, int, const SbRange*);
template void run_clpf_stats<uint8_t>(const ClpfJob<uint8_t>*, const ClpfJob<uint8_t>*, int);
template void run_clpf_stats<uint16_t>(const ClpfJob<uint16_t>*, const ClpfJob<uint16_t>*, int);
template void run_clpf_apply<uint8_t>(const ClpfJob<uint8_t>*, const ClpfJob<uint8_t>*, int);
template void run_clpf_apply<uint16_t>(const ClpfJob<uint16_t>*, const ClpfJob<uint16_t>*, int);
template void run_deblock<uint16_t>(const FrameJob<uint16_t>*, const FrameJob<uint16_t>*, int);
template void run_make_ref<uint16_t>(const FrameJob<uint16_t>*, const Plane3<uint16_t>*, int);
template void run_cdef<uint16_t>(const CdefJob<uint16_t>*, const CdefJob<uint16_t>*, int);
}  // namespace backend
}  // namespace tk

// ---------------------------------------------------------------------------------------------
// C ABI - sequence API
// ---------------------------------------------------------------------------------------------
using namespace tk;

template <typename PIX> struct EncT {
  Engine<PIX> eng;
  std::vector<char> pending;  // thor_hip_next_frame already scheduled the stream's next frame
  std::vector<std::vector<DevFrame<PIX>>> staged;  // [stream][slot]
};
struct thor_hip_encoder {
  SeqParams sp;
  int S = 0;
  bool hbd = false;          // samples are uint16_t (bitdepth > 8)
  EncT<uint8_t>* e8 = nullptr;
  EncT<uint16_t>* e16 = nullptr;
};
#define ENC_DISPATCH(e, body)                                   \
  do {                                                          \
    if ((e)->hbd) { auto& E = *(e)->e16; typedef uint16_t PIXT; body; } \
    else { auto& E = *(e)->e8; typedef uint8_t PIXT; body; }            \
  } while (0)

static SeqParams to_seq(const thor_hip_params& p) {
  SeqParams s;
  s.width = p.width; s.height = p.height; s.qp = p.qp; s.bitdepth = p.bitdepth; s.input_bitdepth = p.input_bitdepth;
  s.frame_rate = p.frame_rate; s.lambda_coeffI = p.lambda_coeffI; s.lambda_coeffP = p.lambda_coeffP;
  s.early_skip_thr = p.early_skip_thr; s.enable_tb_split = p.enable_tb_split; s.enable_pb_split = p.enable_pb_split;
  s.max_num_ref = p.max_num_ref; s.HQperiod = p.HQperiod; s.num_reorder_pics = p.num_reorder_pics; s.interp_ref = p.interp_ref;
  s.dqpP = p.dqpP; s.dqpI = p.dqpI; s.mqpP = p.mqpP; s.intra_period = p.intra_period; s.intra_rdo = p.intra_rdo;
  s.encoder_speed = p.encoder_speed; s.deblocking = p.deblocking; s.cdef = p.cdef; s.clpf = p.clpf;
  s.use_block_contexts = p.use_block_contexts; s.enable_bipred = p.enable_bipred; s.cfl_intra = p.cfl_intra; s.cfl_inter = p.cfl_inter;
  s.dyadic_coding = p.dyadic_coding; s.max_clpf_strength = p.max_clpf_strength;
  s.lambda_coeffB = p.lambda_coeffB; s.lambda_coeffB0 = p.lambda_coeffB0; s.lambda_coeffB1 = p.lambda_coeffB1;
  s.lambda_coeffB2 = p.lambda_coeffB2; s.lambda_coeffB3 = p.lambda_coeffB3;
  s.dqpB = p.dqpB; s.dqpB0 = p.dqpB0; s.dqpB1 = p.dqpB1; s.dqpB2 = p.dqpB2; s.dqpB3 = p.dqpB3;
  s.mqpB = p.mqpB; s.mqpB0 = p.mqpB0; s.mqpB1 = p.mqpB1; s.mqpB2 = p.mqpB2; s.mqpB3 = p.mqpB3;
  return s;
}
static void from_seq(thor_hip_params* p, const SeqParams& s) {
  p->width = s.width; p->height = s.height; p->qp = s.qp; p->bitdepth = s.bitdepth; p->input_bitdepth = s.input_bitdepth;
  p->frame_rate = s.frame_rate; p->lambda_coeffI = s.lambda_coeffI; p->lambda_coeffP = s.lambda_coeffP;
  p->early_skip_thr = s.early_skip_thr; p->enable_tb_split = s.enable_tb_split; p->enable_pb_split = s.enable_pb_split;
  p->max_num_ref = s.max_num_ref; p->HQperiod = s.HQperiod; p->num_reorder_pics = s.num_reorder_pics; p->interp_ref = s.interp_ref;
  p->dqpP = s.dqpP; p->dqpI = s.dqpI; p->mqpP = s.mqpP; p->intra_period = s.intra_period; p->intra_rdo = s.intra_rdo;
  p->encoder_speed = s.encoder_speed; p->deblocking = s.deblocking; p->cdef = s.cdef; p->clpf = s.clpf;
  p->use_block_contexts = s.use_block_contexts; p->enable_bipred = s.enable_bipred; p->cfl_intra = s.cfl_intra; p->cfl_inter = s.cfl_inter;
  p->dyadic_coding = s.dyadic_coding; p->max_clpf_strength = s.max_clpf_strength;
  p->lambda_coeffB = s.lambda_coeffB; p->lambda_coeffB0 = s.lambda_coeffB0; p->lambda_coeffB1 = s.lambda_coeffB1;
  p->lambda_coeffB2 = s.lambda_coeffB2; p->lambda_coeffB3 = s.lambda_coeffB3;
  p->dqpB = s.dqpB; p->dqpB0 = s.dqpB0; p->dqpB1 = s.dqpB1; p->dqpB2 = s.dqpB2; p->dqpB3 = s.dqpB3;
  p->mqpB = s.mqpB; p->mqpB0 = s.mqpB0; p->mqpB1 = s.mqpB1; p->mqpB2 = s.mqpB2; p->mqpB3 = s.mqpB3;
}

static int unsupported(const SeqParams& s) {
  // This path implements the high-efficiency low-delay operating point family; reject the rest
  // loudly rather than silently producing a different stream.
  if (s.bitdepth != s.input_bitdepth || (s.bitdepth != 8 && s.bitdepth != 10 && s.bitdepth != 12))
    return fprintf(stderr, "thor_hip: need bitdepth == input_bitdepth in {8, 10, 12}\n"), 1;
  if (s.num_reorder_pics != 0 && !s.dyadic_coding) return fprintf(stderr, "thor_hip: non-dyadic frame reordering is not implemented\n"), 1;
  if (s.num_reorder_pics < 0 || s.num_reorder_pics > 15 || (s.num_reorder_pics & (s.num_reorder_pics + 1)))
    return fprintf(stderr, "thor_hip: num_reorder_pics must be 0, 1, 3, 7 or 15\n"), 1;
  if (s.interp_ref != 0 && s.interp_ref != 1) return fprintf(stderr, "thor_hip: interp_ref must be 0 or 1\n"), 1;
  if (s.encoder_speed < 0 || s.encoder_speed > 2) return fprintf(stderr, "thor_hip: encoder_speed must be 0, 1 or 2\n"), 1;
  if (s.width % 8 || s.height % 8 || s.width < 16 || s.height < 16) return fprintf(stderr, "thor_hip: bad geometry\n"), 1;
  if (s.max_num_ref < 1 || s.max_num_ref > 4) return fprintf(stderr, "thor_hip: max_num_ref out of range\n"), 1;
  // remaining guards of check_parameters (enc/strings.c:470-555) that matter without rate control / qmtx
  if (s.HQperiod < 1 || s.HQperiod >= 33) return fprintf(stderr, "thor_hip: HQperiod must be in 1..32\n"), 1;
  if (s.num_reorder_pics > 0 && s.HQperiod > 1 && (s.HQperiod % (s.num_reorder_pics + 1)) != 0)
    return fprintf(stderr, "thor_hip: sub-GOP length (num_reorder_pics+1) must divide HQperiod\n"), 1;
  if (s.num_reorder_pics > 0 && s.max_num_ref < 2) return fprintf(stderr, "thor_hip: reordered pictures need more than one reference frame\n"), 1;
  if (s.intra_period < 0 || (s.intra_period % (s.num_reorder_pics + 1)) != 0)
    return fprintf(stderr, "thor_hip: intra_period must be a multiple of the sub-GOP size\n"), 1;
  if (s.qp < 0 || s.qp > 51) return fprintf(stderr, "thor_hip: qp out of range\n"), 1;
  if (s.cdef < 0 || s.cdef > 3 || s.clpf < 0 || s.clpf > 2) return fprintf(stderr, "thor_hip: cdef / clpf out of range\n"), 1;
  if (s.log2_sb_size != 7) return fprintf(stderr, "thor_hip: only 128x128 superblocks are implemented\n"), 1;
  return 0;
}

extern "C" {

int thor_hip_params_from_config(thor_hip_params* p, const char* cfg_path) {
  CliArgs a;
  a.sp.width = 1920; a.sp.height = 1080; a.sp.frame_rate = 60.f;  // enc/strings.c defaults
  if (cfg_path) {
    std::vector<std::string> t = {"-cf", cfg_path};
    cli_apply(a, t);
  }
  from_seq(p, a.sp);
  if (!a.unknown.empty()) return fprintf(stderr, "thor_hip: unknown option %s in %s\n", a.unknown.c_str(), cfg_path), 1;
  if (!a.unsupported.empty()) return fprintf(stderr, "thor_hip: %s (in %s) is not implemented by this path\n", a.unsupported.c_str(), cfg_path), 2;
  return 0;
}

int thor_hip_params_set(thor_hip_params* p, const char* name, const char* value) {
  if (!p || !name || !value) return 1;
  CliArgs a;
  a.sp = to_seq(*p);
  std::vector<std::string> t = {name, value};
  cli_apply(a, t);
  from_seq(p, a.sp);
  if (!a.unknown.empty()) return 1;      // not an option of the reference's table
  if (!a.unsupported.empty()) return 2;  // known, but this value is not implemented (qmtx, rate control, 4:4:4 ...)
  if (!a.infile.empty() || !a.outfile.empty() || !a.recfile.empty() || a.num_frames != 600 || a.skip != 0 || a.streams != 1) return 3;  // front-end option, not an encoder parameter
  return 0;
}

int thor_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

thor_hip_encoder* thor_hip_open(const thor_hip_params* p, int num_streams, int device) {
  if (!p || num_streams < 1) return nullptr;
  SeqParams s = to_seq(*p);
  if (unsupported(s)) return nullptr;
  if (!ensure_init(device)) return nullptr;
  thor_hip_encoder* e = new thor_hip_encoder;
  e->sp = s;
  e->S = num_streams;
  e->hbd = s.bitdepth > 8;
  if (e->hbd) e->e16 = new EncT<uint16_t>; else e->e8 = new EncT<uint8_t>;
  ENC_DISPATCH(e, { E.eng.open(s, num_streams); E.staged.resize(num_streams); E.pending.assign(num_streams, 0); });
  return e;
}

void thor_hip_close(thor_hip_encoder* e) {
  if (!e) return;
  ENC_DISPATCH(e, {
    for (auto& v : E.staged)
      for (auto& f : v)
        if (f.base_y) f.release();
    E.eng.close();
  });
  delete e->e8;
  delete e->e16;
  delete e;
}

int thor_hip_begin_sequence(thor_hip_encoder* e, int stream, int skip, int num_frames, int file_frames) {
  if (!e || stream < 0 || stream >= e->S || skip < 0 || num_frames < 1 || file_frames < skip + num_frames) return 1;
  ENC_DISPATCH(e, { E.eng.begin_sequence(stream, skip, num_frames, file_frames); E.pending[stream] = 0; });
  return 0;
}

int thor_hip_next_frame(thor_hip_encoder* e, int stream, int* display_index) {
  if (!e || stream < 0 || stream >= e->S) return 0;
  int ok = 0;
  ENC_DISPATCH(e, {
    if (E.pending[stream]) ok = 1;
    else ok = E.eng.schedule(stream) ? 1 : 0;
    E.pending[stream] = (char)ok;
    if (ok && display_index) *display_index = E.eng.st[stream].cur.frame_num;
  });
  return ok;
}

int thor_hip_stage_frame(thor_hip_encoder* e, int stream, int slot, const void* yuv) {
  if (!e || stream < 0 || stream >= e->S || slot < 0 || !yuv) return 1;
  ENC_DISPATCH(e, {
    auto& v = E.staged[stream];
    if ((int)v.size() <= slot) v.resize(slot + 1);
    if (!v[slot].base_y) v[slot].alloc(e->sp.width, e->sp.height, 0);
    DevFrame<PIXT> keep = E.eng.st[stream].orig;
    E.eng.st[stream].orig = v[slot];
    E.eng.upload_orig(stream, (const PIXT*)yuv);
    E.eng.st[stream].orig = keep;
  });
  return 0;
}

// Same as thor_hip_stage_frame for a frame that already lives in HBM (e.g. a torch CUDA tensor the caller derived from a
// clip broadcast over RCCL): three device-to-device 2-D copies on the library's stream; the source may be released
// when the call returns.
int thor_hip_stage_frame_device(thor_hip_encoder* e, int stream, int slot, const void* dev_yuv) {
  if (!e || stream < 0 || stream >= e->S || slot < 0 || !dev_yuv) return 1;
  ENC_DISPATCH(e, {
    auto& v = E.staged[stream];
    if ((int)v.size() <= slot) v.resize(slot + 1);
    if (!v[slot].base_y) v[slot].alloc(e->sp.width, e->sp.height, 0);
    const int w = e->sp.width;
    const int h = e->sp.height;
    const PIXT* src = (const PIXT*)dev_yuv;
    const Plane3<PIXT>& d = v[slot].p;
    HIPCHECK(hipMemcpy2DAsync(d.y, (size_t)d.sy * sizeof(PIXT), src, (size_t)w * sizeof(PIXT), (size_t)w * sizeof(PIXT), h, hipMemcpyDeviceToDevice, g_stream));
    src += (size_t)w * h;
    HIPCHECK(hipMemcpy2DAsync(d.u, (size_t)d.sc * sizeof(PIXT), src, (size_t)(w / 2) * sizeof(PIXT), (size_t)(w / 2) * sizeof(PIXT), h / 2, hipMemcpyDeviceToDevice, g_stream));
    src += (size_t)(w / 2) * (h / 2);
    HIPCHECK(hipMemcpy2DAsync(d.v, (size_t)d.sc * sizeof(PIXT), src, (size_t)(w / 2) * sizeof(PIXT), (size_t)(w / 2) * sizeof(PIXT), h / 2, hipMemcpyDeviceToDevice, g_stream));
    HIPCHECK(hipStreamSynchronize(g_stream));
  });
  return 0;
}

int thor_hip_encode_staged(thor_hip_encoder* e, const int* slots) {
  if (!e || !slots) return 1;
  int rc = 0;
  ENC_DISPATCH(e, {
    std::vector<DevFrame<PIXT>> keep(e->S);
    std::vector<FrameParams> fp(e->S);
    for (int s = 0; s < e->S && !rc; s++)
      if (slots[s] < 0 || slots[s] >= (int)E.staged[s].size() || !E.staged[s][slots[s]].base_y) rc = 2;
    if (!rc) {
      for (int s = 0; s < e->S; s++) {
        keep[s] = E.eng.st[s].orig;
        E.eng.st[s].orig = E.staged[s][slots[s]];
        if (!E.pending[s] && !E.eng.schedule(s)) { fprintf(stderr, "thor_hip: stream %d has no frame left to code\n", s); abort(); }
        E.pending[s] = 0;
        fp[s] = E.eng.st[s].cur;
      }
      E.eng.encode_frames(fp);
      for (int s = 0; s < e->S; s++) E.eng.st[s].orig = keep[s];
    }
  });
  return rc;
}

int thor_hip_encode_staged_run(thor_hip_encoder* e, int nframes, thor_hip_frames_done_fn done, void* user) {
  if (!e || nframes < 0) return 1;
  int rc = 0;
  ENC_DISPATCH(e, {
    // Validate BEFORE the first launch (a failure inside encode_run would leave half-frames in flight): dry-run every stream's coding-order
    // schedule on a copy - the sequence of display indices does not depend on the reference ring - and check that each of the next `nframes`
    // frames exists (rc 2) and is staged (rc 3).  Nothing is touched when the run is refused.
    for (int s = 0; s < e->S && !rc; s++) {
      GopScheduler g = E.eng.st[s].gop;
      if (!g.started) g.init(E.eng.sp, 0, 1 << 28, 1 << 28);
      for (int f = 0; f < nframes && !rc; f++) {
        FrameParams fpar;
        int abs_frame = 0;
        if (f == 0 && E.pending[s]) fpar = E.eng.st[s].cur;   // already scheduled by thor_hip_next_frame
        else if (!g.next(fpar, abs_frame, [&](int idx) { return E.eng.st[s].ring[idx].frame_num; })) { rc = 2; break; }
        g.advance(fpar);   // the engine advances the schedule when the frame is finished (tk_encoder.h:finish_frames)
        const int slot = fpar.frame_num;
        if (slot < 0 || slot >= (int)E.staged[s].size() || !E.staged[s][slot].base_y) {
          fprintf(stderr, "thor_hip: stream %d: frame %d is not staged\n", s, slot);
          rc = 3;
        }
      }
    }
    if (rc) return rc;
    std::vector<DevFrame<PIXT>> keep(e->S);
    for (int s = 0; s < e->S; s++) keep[s] = E.eng.st[s].orig;
    E.eng.encode_run(nframes,
        [&](int s) -> bool {
          if (!E.pending[s] && !E.eng.schedule(s)) { rc = 2; return false; }
          E.pending[s] = 0;
          const int slot = E.eng.st[s].cur.frame_num;
          if (slot < 0 || slot >= (int)E.staged[s].size() || !E.staged[s][slot].base_y) {
            fprintf(stderr, "thor_hip: stream %d: frame %d is not staged\n", s, slot);
            rc = 3;
            return false;
          }
          E.eng.st[s].orig = E.staged[s][slot];
          return true;
        },
        [&](int first, int count) { if (done) done(user, first, count); });
    for (int s = 0; s < e->S; s++) E.eng.st[s].orig = keep[s];
  });
  return rc;
}
int thor_hip_last_display_index(const thor_hip_encoder* e, int stream) {
  if (!e || stream < 0 || stream >= e->S) return -1;
  const int n = e->hbd ? e->e16->eng.st[stream].num_encoded : e->e8->eng.st[stream].num_encoded;
  if (n < 1) return -1;
  return e->hbd ? e->e16->eng.st[stream].cur.frame_num : e->e8->eng.st[stream].cur.frame_num;
}

int thor_hip_encode_frame(thor_hip_encoder* e, const void* const* yuv) {
  if (!e || !yuv) return 1;
  ENC_DISPATCH(e, {
    std::vector<FrameParams> fp(e->S);
    for (int s = 0; s < e->S; s++) {
      E.eng.upload_orig(s, (const PIXT*)yuv[s]);
      if (!E.pending[s] && !E.eng.schedule(s)) { fprintf(stderr, "thor_hip: stream %d has no frame left to code\n", s); abort(); }
      E.pending[s] = 0;
      fp[s] = E.eng.st[s].cur;
    }
    E.eng.encode_frames(fp);
  });
  return 0;
}

size_t thor_hip_stream_bytes(const thor_hip_encoder* e, int stream) {
  if (!e || stream < 0 || stream >= e->S) return 0;
  return e->hbd ? e->e16->eng.st[stream].out.size() : e->e8->eng.st[stream].out.size();
}
const uint8_t* thor_hip_stream_data(const thor_hip_encoder* e, int stream) {
  if (!e || stream < 0 || stream >= e->S) return nullptr;
  return e->hbd ? e->e16->eng.st[stream].out.data() : e->e8->eng.st[stream].out.data();
}
int thor_hip_get_recon(thor_hip_encoder* e, int stream, void* yuv_out) {
  if (!e || stream < 0 || stream >= e->S || !yuv_out) return 1;
  ENC_DISPATCH(e, { E.eng.download_rec(stream, (PIXT*)yuv_out); });
  return 0;
}
void thor_hip_kernel_time(thor_hip_encoder*, double* sb_ms, long* sb_launches, double* filter_ms) {
  if (sb_ms) *sb_ms = g_clk.sb_ms;
  if (sb_launches) *sb_launches = g_clk.sb_launches;
  if (filter_ms) *filter_ms = g_clk.filt_ms;
}
void thor_hip_read_prof(thor_hip_encoder* e, long long out[32]) { if (!e || !out) return; ENC_DISPATCH(e, { backend::d2h(out, E.eng.d_prof, 32 * sizeof(long long)); }); }
void thor_hip_kernel_time_reset(thor_hip_encoder*) { g_clk.sb_ms = g_clk.filt_ms = 0; g_clk.sb_launches = 0; }
void thor_hip_read_stats(thor_hip_encoder* e, unsigned long long out[4], int reset) {
  if (!e || !out) return;
  ENC_DISPATCH(e, { backend::d2h(out, E.eng.d_stats, 4 * sizeof(unsigned long long)); if (reset) backend::dev_memset(E.eng.d_stats, 0, 8 * sizeof(unsigned long long)); });
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// C ABI - drop-in seam (enc/encode_frame.h:32-33)
// ---------------------------------------------------------------------------------------------
#include "../../include/thor_abi.h"
#include <map>

static void seam_fatal(const char* msg) {  // fatalerror() convention, common/global.h:38-44
  fprintf(stderr, "Run-time error...\n%s\n...now exiting to system...\n", msg);
  abort();
}

// putbits(n, val) of enc/putbits.c:109-128 for 1 <= n <= 16 (same lazy flush: a full accumulator is only written out
// by the next put, so the caller's (bitbuf, bitrest, bytepos) end up exactly as if the reference had written the bits)
static void stream_put(thor_stream* s, unsigned n, unsigned val) {
  val &= (1u << n) - 1u;
  if (n <= s->bitrest) {
    s->bitbuf |= val << (s->bitrest - n);
    s->bitrest -= n;
  } else {
    const unsigned rest = n - s->bitrest;
    s->bitbuf |= val >> rest;
    if (s->bytepos + 4 > s->bytesize) seam_fatal("Run out of bits in stream buffer.");
    for (int i = 3; i >= 0; --i) s->bitstream[s->bytepos++] = (uint8_t)((s->bitbuf >> (8 * i)) & 0xff);
    s->bitbuf = (val & ((1u << rest) - 1u)) << (32 - rest);
    s->bitrest = 32 - rest;
  }
}

template <typename PIX> struct SeamState {
  Engine<PIX> eng;
};
template <typename PIX> static std::map<const void*, SeamState<PIX>*>& seams() {
  static std::map<const void*, SeamState<PIX>*> m;
  return m;
}

template <typename PIX> static void encode_frame_impl(struct thor_encoder_info* ei) {
  if (!ei || !ei->params || !ei->orig || !ei->rec || !ei->stream) seam_fatal("encode_frame: null encoder_info member");
  if ((ei->params->bitdepth > 8) != (sizeof(PIX) == 2)) seam_fatal("thor_hip: frame sample size does not match params->bitdepth");
  const thor_enc_params& ep = *ei->params;
  thor_frame_info& fi = ei->frame_info;
  SeamState<PIX>*& st = seams<PIX>()[ei];
  if (!st) {
    SeqParams s;
    s.width = ei->width; s.height = ei->height; s.qp = (int)ep.qp; s.bitdepth = ep.bitdepth; s.input_bitdepth = ep.input_bitdepth;
    s.frame_rate = ep.frame_rate; s.lambda_coeffI = ep.lambda_coeffI; s.lambda_coeffP = ep.lambda_coeffP;
    s.early_skip_thr = ep.early_skip_thr; s.enable_tb_split = ep.enable_tb_split; s.enable_pb_split = ep.enable_pb_split;
    s.max_num_ref = ep.max_num_ref; s.HQperiod = THOR_MAX_REF_FRAMES - 1;  // window large enough for any ref_array the caller builds
    s.num_reorder_pics = ep.num_reorder_pics; s.interp_ref = ep.interp_ref; s.dqpP = ep.dqpP; s.dqpI = ep.dqpI; s.mqpP = ep.mqpP;
    s.intra_period = ep.intra_period; s.intra_rdo = ep.intra_rdo; s.encoder_speed = ep.encoder_speed; s.deblocking = ep.deblocking;
    s.cdef = ep.cdef; s.clpf = ep.clpf; s.use_block_contexts = ep.use_block_contexts; s.enable_bipred = ep.enable_bipred;
    s.cfl_intra = ep.cfl_intra; s.cfl_inter = ep.cfl_inter; s.log2_sb_size = ep.log2_sb_size;
    s.max_clpf_strength = ep.max_clpf_strength;
    s.dyadic_coding = 1;  // the caller owns the GOP structure; only the window size matters here
    if (ep.subsample != 420 || ep.log2_sb_size != 7 || ep.qmtx || ep.max_delta_qp || ep.bitrate || ep.sync)
      seam_fatal("thor_hip: unsupported encoder parameters (need 4:2:0, 128x128 SB, no qmtx / delta-QP / rate control / sync)");
    if (unsupported(s)) seam_fatal("thor_hip: unsupported encoder parameters");
    if (!ensure_init(getenv("THOR_HIP_DEVICE") ? atoi(getenv("THOR_HIP_DEVICE")) : 0)) seam_fatal("thor_hip: HIP device not usable");
    st = new SeamState<PIX>;
    st->eng.raw_frames = true;
    st->eng.external_interp = true;  // the caller interpolates (enc/mainenc.c:353) and hands the frame over
    st->eng.open(s, 1);
  }
  Engine<PIX>& eng = st->eng;
  if (fi.interp_ref > 1) seam_fatal("thor_hip: interp_ref > 1 is not implemented");
  if (fi.num_ref > kMaxRefs) seam_fatal("thor_hip: more than 4 references");
  FrameParams f;
  f.frame_type = fi.frame_type; f.qp = fi.qp; f.num_ref = fi.num_ref; f.frame_num = fi.frame_num; f.interp_ref = fi.interp_ref;
  f.num_intra_modes = fi.num_intra_modes; f.b_level = fi.b_level;
  for (int r = 0; r < fi.num_ref; r++) {
    if (fi.ref_array[r] < -1 || fi.ref_array[r] >= eng.ring_size) seam_fatal("thor_hip: reference index outside the device window");
    f.ref_array[r] = fi.ref_array[r];
    if (fi.ref_array[r] == -1) {  // interpolated frame built by the caller
      if (!ei->interp_frames[0] || !ep.interp_ref) seam_fatal("thor_hip: ref_array -1 without an interpolated frame");
      const thor_yuv_frame& q = *ei->interp_frames[0];
      DevFrame<PIX>& g = eng.st[0].interp;
      auto push = [&](const PIX* hp, int hs, PIX* dp, int ds, int w, int h, int padw, int padh) {
        std::vector<PIX> buf((size_t)(h + 2 * padh) * ds);
        for (int i = -padh; i < h + padh; i++) memcpy(&buf[(size_t)(i + padh) * ds], hp + (ptrdiff_t)i * hs - padw, (w + 2 * padw) * sizeof(PIX));
        backend::h2d(dp - (size_t)padh * ds - padw, buf.data(), (buf.size() - (size_t)(ds - (w + 2 * padw))) * sizeof(PIX));
      };
      if (q.pad_hor_y < kPadY || q.pad_ver_y < kPadY) seam_fatal("thor_hip: interpolated frame padding too small");
      push((const PIX*)q.y, q.stride_y, g.p.y, g.p.sy, ei->width, ei->height, kPadY, kPadY);
      push((const PIX*)q.u, q.stride_c, g.p.u, g.p.sc, ei->width / 2, ei->height / 2, kPadY / 2, kPadY / 2);
      push((const PIX*)q.v, q.stride_c, g.p.v, g.p.sc, ei->width / 2, ei->height / 2, kPadY / 2, kPadY / 2);
      g.frame_num = q.frame_num;
    }
  }
  // lambda_coeff by frame type / B level (enc/encode_frame.c:655-672)
  if (fi.frame_type == F_I) f.lambda_coeff = ep.lambda_coeffI;
  else if (fi.frame_type == F_P) f.lambda_coeff = ep.lambda_coeffP;
  else f.lambda_coeff = fi.b_level == 0 ? ep.lambda_coeffB0 : fi.b_level == 1 ? ep.lambda_coeffB1 : fi.b_level == 2 ? ep.lambda_coeffB2
                                          : fi.b_level == 3 ? ep.lambda_coeffB3 : ep.lambda_coeffB;
  fi.lambda_coeff = f.lambda_coeff;
  fi.lambda = f.lambda_coeff * kSquaredLambdaQP[f.qp];
  fi.prev_qp = fi.qp;
  const thor_yuv_frame& o = *ei->orig;
  eng.upload_planes(0, (const PIX*)o.y, o.stride_y, (const PIX*)o.u, (const PIX*)o.v, o.stride_c);
  eng.st[0].num_encoded = fi.frame_num;  // only used for bookkeeping
  eng.st[0].bit_phase = (8 * (int)ei->stream->bytepos + (32 - (int)ei->stream->bitrest)) & 31;  // get_bit_pos() of the caller's stream
  std::vector<FrameParams> fp(1, f);
  eng.encode_frames(fp);
  // bits -> caller's stream (the caller flushes: enc/mainenc.c:595)
  HostBits& b = eng.st[0].bits;
  {
    int i = 0;
    for (; i + 16 <= b.nbits; i += 16) stream_put(ei->stream, 16, (b.w[i >> 5] >> (16 - (i & 16))) & 0xffffu);
    for (; i < b.nbits; i++) stream_put(ei->stream, 1, (unsigned)b.get(i));
  }
  b.clear();
  // reconstruction -> caller's rec frame
  {
    thor_yuv_frame& r = *ei->rec;
    std::vector<PIX> tmp((size_t)ei->width * ei->height * 3 / 2);
    eng.download_rec(0, tmp.data());
    const int w = ei->width, h = ei->height;
    for (int i = 0; i < h; i++) memcpy((PIX*)r.y + (size_t)i * r.stride_y, &tmp[(size_t)i * w], w * sizeof(PIX));
    const PIX* cu = &tmp[(size_t)w * h]; const PIX* cv = cu + (size_t)(w / 2) * (h / 2);
    for (int i = 0; i < h / 2; i++) {
      memcpy((PIX*)r.u + (size_t)i * r.stride_c, cu + (size_t)i * (w / 2), (w / 2) * sizeof(PIX));
      memcpy((PIX*)r.v + (size_t)i * r.stride_c, cv + (size_t)i * (w / 2), (w / 2) * sizeof(PIX));
    }
  }
  // deblock_data[] as copy_deblock_data leaves it (enc/encode_block.c:1568-1613): the device keeps it as 16-byte DbCells
  if (ei->deblock_data) {
    std::vector<DbCell> cells(eng.num_cells());
    eng.download_cells(0, cells.data());
    for (size_t i = 0; i < cells.size(); i++) {
      const DdFields c = dd_fields(cells[i]);
      thor_deblock_data& d = ei->deblock_data[i];
      d.mode = c.mode; d.cbp_y = c.cbp_y; d.cbp_u = c.cbp_u; d.cbp_v = c.cbp_v;
      d.size = (uint8_t)c.size; d.tb_split = (uint8_t)c.tb_split; d.pb_part = c.pb_part;
      d.inter_pred.mv0.x = (int16_t)c.mv0x; d.inter_pred.mv0.y = (int16_t)c.mv0y;
      d.inter_pred.mv1.x = (int16_t)c.mv1x; d.inter_pred.mv1.y = (int16_t)c.mv1y;
      d.inter_pred.ref_idx0 = (uint32_t)c.ref_idx0; d.inter_pred.ref_idx1 = (uint32_t)c.ref_idx1; d.inter_pred.bipred_flag = (uint32_t)c.bipred_flag;
    }
  }
  // sliding window of the caller's reference pointers + padded copy (enc/encode_frame.c:826-835)
  {
    thor_yuv_frame* last = ei->ref[THOR_MAX_REF_FRAMES - 1];
    memmove(ei->ref + 1, ei->ref, sizeof(thor_yuv_frame*) * (THOR_MAX_REF_FRAMES - 1));
    ei->ref[0] = last;
    thor_yuv_frame& d = *ei->ref[0];
    const DevFrame<PIX>& g = eng.st[0].ring[0];
    d.frame_num = ei->rec->frame_num;
    const int ph = d.pad_ver_y, pw = d.pad_hor_y, pch = d.pad_ver_c, pcw = d.pad_hor_c;
    auto pull = [&](PIX* hp, int hs, const PIX* dp, int ds, int w, int h, int padw, int padh) {
      std::vector<PIX> buf((size_t)(h + 2 * padh) * ds);
      backend::d2h(buf.data(), dp - (size_t)padh * ds - padw, (buf.size() - (size_t)(ds - (w + 2 * padw))) * sizeof(PIX));
      for (int i = -padh; i < h + padh; i++)
        memcpy(hp + (ptrdiff_t)i * hs - padw, &buf[(size_t)(i + padh) * ds], (w + 2 * padw) * sizeof(PIX));
    };
    pull((PIX*)d.y, d.stride_y, g.p.y, g.p.sy, ei->width, ei->height, pw < kPadY ? pw : kPadY, ph < kPadY ? ph : kPadY);
    pull((PIX*)d.u, d.stride_c, g.p.u, g.p.sc, ei->width / 2, ei->height / 2, pcw < kPadY / 2 ? pcw : kPadY / 2, pch < kPadY / 2 ? pch : kPadY / 2);
    pull((PIX*)d.v, d.stride_c, g.p.v, g.p.sc, ei->width / 2, ei->height / 2, pcw < kPadY / 2 ? pcw : kPadY / 2, pch < kPadY / 2 ? pch : kPadY / 2);
  }
  ei->cdef_damping = 5;
}

extern "C" void encode_frame_lbd(struct thor_encoder_info* ei) { encode_frame_impl<uint8_t>(ei); }
extern "C" void encode_frame_hbd(struct thor_encoder_info* ei) { encode_frame_impl<uint16_t>(ei); }

// ---------------------------------------------------------------------------------------------
// C ABI - kernel-level batch entry points (known-answer tests)
// ---------------------------------------------------------------------------------------------
namespace tk {
// The kernels behind the known-answer entry points run the product's device code on one block / transform unit per workgroup of
// one wavefront; PIX = uint8_t (the reference's _lbd functions) or uint16_t (_hbd, bitdepth 9..12).
template <typename PIX>
__global__ __launch_bounds__(64) void k_kat_sad(const PIX* org, int w, int h, const PIX* refp, int rstride, int bx, int by, const int* cand, int n,
                                               uint32_t* out) {
  // the product's full-pel evaluator (tk_me.h:seg_sads, row segment per lane), plane reads only (no search window)
  const Team t = mk_team((int)threadIdx.x, 64);
  struct KC { const PIX* p; int dx, dy; };
  MeWin win;
  win.on = 0; win.w32 = nullptr; win.ox = win.oy = win.Ww = win.Wh = win.pitch = 0;
  auto cnd = [&](int c) -> KC {
    KC x;
    x.dx = cand[2 * c]; x.dy = cand[2 * c + 1];
    x.p = refp + (size_t)(by + x.dy) * rstride + bx + x.dx;
    return x;
  };
  seg_sads<SP_GLOBAL>(t, n, org, w, rstride, w, h, win, cnd, [&](int c, const KC&, int sad, int mine) { if (mine) out[c] = (uint32_t)sad; });
}
template <typename PIX>
__global__ __launch_bounds__(64) void k_kat_interp(const PIX* ref0, int rstride, int pic_w, int pic_h, int bx, int by, int w, int h,
                                                  const int16_t* mv, int bipred, int bitdepth, PIX* out) {
  const Team t = mk_team((int)threadIdx.x, 64);
  const int i = blockIdx.x;
  pred_luma<SP_GLOBAL>(t, out + (size_t)i * w * h, w, ref0 + (size_t)by * rstride + bx, rstride, w, h, mk_mv(mv[2 * i], mv[2 * i + 1]), 0,
            bipred, pic_w, pic_h, bx, by, bitdepth);
}
template <typename PIX>
__global__ __launch_bounds__(64) void k_kat_tu(const PIX* org, const PIX* pred, int size, int qp, int coeff_type, int fast, int bitdepth,
                                              int16_t* coefq, PIX* rec, int* cbp) {
  __shared__ XformWs xf;
  __shared__ XformTabs tabs;
  __shared__ int16_t cq[256];
  const Team t = mk_team((int)threadIdx.x, 64, tabs.izz);
  xf.prof = nullptr;
  xf.tabs = &tabs;
  xform_tables_fill(&tabs, (int)threadIdx.x, 64);
  t.sync();
  const int i = blockIdx.x, qs = size < 16 ? size : 16;
  const size_t o = (size_t)i * size * size;
  int c = code_tu(t, &xf, org + o, size, pred + o, size, rec + o, size, size, qp, coeff_type, fast, cq, bitdepth);
  for (int k = threadIdx.x; k < qs * qs; k += 64) coefq[(size_t)i * qs * qs + k] = cq[k];
  if (threadIdx.x == 0) cbp[i] = c;
}

// ---- round 6: known-answer kernels for the sample kernels that were only covered by whole-stream hashes -------------------------
// One wavefront per item, running exactly the device functions the encoder calls.
template <typename PIX>
__global__ __launch_bounds__(64) void k_kat_intra(const PIX* plane, int stride, int bitdepth, int size, int tb_split, const int* par, const PIX* rblocks,
                                                 PIX* out) {
  __shared__ IntraEdge<PIX> edge;
  const Team t = mk_team((int)threadIdx.x, 64);
  const int it = blockIdx.x;
  const int* q = par + 7 * it;   // ypos, xpos (coding block), upright, downleft, mode, i, j (transform unit inside the block)
  const int cbs = tb_split ? 2 * size : size;
  const PIX* rblock = tb_split ? rblocks + (size_t)it * cbs * cbs + q[5] * cbs + q[6] : nullptr;
  make_edges<SP_GLOBAL>(t, &edge, plane + (size_t)q[0] * stride + q[1], stride, rblock, cbs, q[5], q[6], q[0], q[1], size, q[2], q[3], tb_split, bitdepth);
  pred_intra<SP_GLOBAL>(t, &edge, q[0] + q[5], q[1] + q[6], size, out + (size_t)it * size * size, size, q[4], bitdepth);
}
// (pred_inter_yuv / improve_uv are __noinline__ functions the superblock kernel calls too: a kernel with a larger register budget calling them would raise
// THEIR budget and with it the superblock kernel's VGPR count - 227 instead of 168, two workgroups per CU instead of three, measured in round 6 - so these two
// kernels carry the superblock kernel's launch bounds)
template <typename PIX>
__global__ __launch_bounds__(kWgThreads, (sizeof(PIX) == 1 ? (int)kOcc : 2)) void k_kat_inter_yuv(Plane3<PIX> ref, int width, int height, int bitdepth, int size, const int* par, const int16_t* mv, PIX* out) {
  const Team t = mk_team((int)threadIdx.x, 64);
  const int it = blockIdx.x;
  const int* q = par + 5 * it;   // ypos, xpos, sign, enable_bipred, split
  mv_t m[4];
  for (int k = 0; k < 4; k++) m[k] = mk_mv(mv[(it * 4 + k) * 2], mv[(it * 4 + k) * 2 + 1]);
  PIX* o = out + (size_t)it * (size * size * 3 / 2);
  pred_inter_yuv<SP_GLOBAL>(t, ref, o, o + size * size, o + size * size * 5 / 4, q[0], q[1], size, size, size, m, q[2], width, height, q[3], q[4], bitdepth);
}
template <typename PIX> __global__ __launch_bounds__(64) void k_kat_average(const PIX* a, const PIX* b, int size, PIX* out) {
  const Team t = mk_team((int)threadIdx.x, 64);
  const size_t o = (size_t)blockIdx.x * (size * size * 3 / 2);
  const int n = size * size, c = n / 4;
  average_yuv<SP_GLOBAL>(t, out + o, out + o + n, out + o + n + c, a + o, a + o + n, a + o + n + c, b + o, b + o + n, b + o + n + c, size, size, size);
}
template <typename PIX>
__global__ __launch_bounds__(kWgThreads, (sizeof(PIX) == 1 ? (int)kOcc : 2)) void k_kat_cfl(const PIX* y, PIX* uv, const PIX* ry, int n, int bitdepth) {
  const Team t = mk_team((int)threadIdx.x, 64);
  const int it = blockIdx.x, c = (n / 2) * (n / 2);
  improve_uv<PIX, SP_GLOBAL>(t, nullptr, y + (size_t)it * n * n, uv + (size_t)it * 2 * c, uv + (size_t)it * 2 * c + c, ry + (size_t)it * n * n, n, n, n, bitdepth);
}
template <typename PIX> __global__ void k_kat_cdef_dir(const PIX* blocks, int n, int cs, int* dir, int* var) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  int v = 0;
  dir[i] = cdef_find_dir(blocks + (size_t)i * 64, 8, &v, cs);
  var[i] = v;
}
template <typename PIX>
__global__ __launch_bounds__(64) void k_kat_cdef_filter(const PIX* plane, int w, int h, int stride, int bsize, int cs, const int* par, PIX* out) {
  const int it = blockIdx.x, k = threadIdx.x;
  if (k >= bsize * bsize) return;
  const int* q = par + 7 * it;   // x0, y0, pri, sec, dir, pri_damping, sec_damping
  const int x = q[0] + k % bsize, y = q[1] + k / bsize;
  out[(size_t)it * bsize * bsize + k] = (PIX)cdef_filter_px(plane, stride, x, y, w, h, q[2], q[3], q[4], q[5], q[6], cs);
}
}  // namespace tk

template <typename T> static T* to_dev(const T* h, size_t n) {
  T* d = (T*)backend::dev_alloc(n * sizeof(T));
  if (h) backend::h2d(d, h, n * sizeof(T));
  return d;
}

template <typename PIX>
static int kat_sad_batch(const PIX* org, int w, int h, const PIX* ref_plane, int plane_w, int plane_h, int rstride, int bx, int by, const int* cand,
                         int n, uint32_t* out) {
  if (!org || !ref_plane || !cand || !out || n <= 0 || w < 4 || h < 4 || (w & (w - 1)) || (h & (h - 1))) return 1;
  for (int i = 0; i < n; i++) {
    int x = bx + cand[2 * i], y = by + cand[2 * i + 1];
    if (x < 0 || y < 0 || x + w > plane_w || y + h > plane_h) return 2;
  }
  if (!ensure_init(g_inited ? g_device : 0)) return 3;
  PIX* d_org = to_dev(org, (size_t)w * h);
  // the evaluator reads whole 16-byte row segments: 16 zeroed samples of slack behind the plane on the device (dev_alloc clears);
  // only the caller's rstride * plane_h samples are read from the host buffer
  PIX* d_ref = to_dev<PIX>(nullptr, (size_t)rstride * plane_h + 16);
  backend::h2d(d_ref, ref_plane, (size_t)rstride * plane_h * sizeof(PIX));
  int* d_c = to_dev(cand, (size_t)2 * n);
  uint32_t* d_o = to_dev<uint32_t>(nullptr, n);
  hipLaunchKernelGGL(k_kat_sad<PIX>, dim3(1), dim3(64), 0, g_stream, d_org, w, h, d_ref, rstride, bx, by, d_c, n, d_o);
  HIPCHECK(hipGetLastError());
  backend::d2h(out, d_o, (size_t)n * 4);
  backend::dev_free(d_org); backend::dev_free(d_ref); backend::dev_free(d_c); backend::dev_free(d_o);
  return 0;
}
extern "C" int thor_hip_sad_batch(const uint8_t* org, int w, int h, const uint8_t* ref_plane, int plane_w, int plane_h, int rstride,
                                  int bx, int by, const int* cand, int n, uint32_t* out) {
  return kat_sad_batch<uint8_t>(org, w, h, ref_plane, plane_w, plane_h, rstride, bx, by, cand, n, out);
}
extern "C" int thor_hip_sad_batch_hbd(const uint16_t* org, int w, int h, const uint16_t* ref_plane, int plane_w, int plane_h, int rstride,
                                      int bx, int by, const int* cand, int n, uint32_t* out) {
  return kat_sad_batch<uint16_t>(org, w, h, ref_plane, plane_w, plane_h, rstride, bx, by, cand, n, out);
}

template <typename PIX>
static int kat_interp_luma(const PIX* ref_plane, int plane_w, int plane_h, int rstride, int pad, int bx, int by, int w, int h, const int16_t* mv,
                           int n, int bipred, int bitdepth, PIX* out) {
  if (!ref_plane || !mv || !out || n <= 0) return 1;
  if (!ensure_init(g_inited ? g_device : 0)) return 3;
  const size_t total = (size_t)rstride * (plane_h + 2 * pad);
  PIX* d_ref = to_dev(ref_plane, total);
  int16_t* d_mv = to_dev(mv, (size_t)2 * n);
  PIX* d_o = to_dev<PIX>(nullptr, (size_t)n * w * h);
  hipLaunchKernelGGL(k_kat_interp<PIX>, dim3(n), dim3(64), 0, g_stream, d_ref + (size_t)pad * rstride + pad, rstride, plane_w, plane_h, bx,
                     by, w, h, d_mv, bipred, bitdepth, d_o);
  HIPCHECK(hipGetLastError());
  backend::d2h(out, d_o, (size_t)n * w * h * sizeof(PIX));
  backend::dev_free(d_ref); backend::dev_free(d_mv); backend::dev_free(d_o);
  return 0;
}
extern "C" int thor_hip_interp_luma(const uint8_t* ref_plane, int plane_w, int plane_h, int rstride, int pad, int bx, int by, int w,
                                    int h, const int16_t* mv, int n, int bipred, uint8_t* out) {
  return kat_interp_luma<uint8_t>(ref_plane, plane_w, plane_h, rstride, pad, bx, by, w, h, mv, n, bipred, 8, out);
}
extern "C" int thor_hip_interp_luma_hbd(const uint16_t* ref_plane, int plane_w, int plane_h, int rstride, int pad, int bx, int by, int w,
                                        int h, const int16_t* mv, int n, int bipred, int bitdepth, uint16_t* out) {
  if (bitdepth < 9 || bitdepth > 12) return 1;
  return kat_interp_luma<uint16_t>(ref_plane, plane_w, plane_h, rstride, pad, bx, by, w, h, mv, n, bipred, bitdepth, out);
}

template <typename PIX>
static int kat_code_tu_batch(const PIX* org, const PIX* pred, int size, int qp, int coeff_type, int fast, int n, int bitdepth, int16_t* coefq,
                             PIX* rec, int* cbp) {
  if (!org || !pred || !coefq || !rec || !cbp || n <= 0) return 1;
  if (size != 4 && size != 8 && size != 16 && size != 32 && size != 64 && size != 128) return 2;
  if (!ensure_init(g_inited ? g_device : 0)) return 3;
  const size_t px = (size_t)n * size * size;
  const int qs = size < 16 ? size : 16;
  PIX* d_org = to_dev(org, px);
  PIX* d_pred = to_dev(pred, px);
  PIX* d_rec = to_dev<PIX>(nullptr, px);
  int16_t* d_cq = to_dev<int16_t>(nullptr, (size_t)n * qs * qs);
  int* d_cbp = to_dev<int>(nullptr, n);
  hipLaunchKernelGGL(k_kat_tu<PIX>, dim3(n), dim3(64), 0, g_stream, d_org, d_pred, size, qp, coeff_type, fast, bitdepth, d_cq, d_rec, d_cbp);
  HIPCHECK(hipGetLastError());
  backend::d2h(coefq, d_cq, (size_t)n * qs * qs * 2);
  backend::d2h(rec, d_rec, px * sizeof(PIX));
  backend::d2h(cbp, d_cbp, (size_t)n * 4);
  backend::dev_free(d_org); backend::dev_free(d_pred); backend::dev_free(d_rec); backend::dev_free(d_cq); backend::dev_free(d_cbp);
  return 0;
}
extern "C" int thor_hip_code_tu_batch(const uint8_t* org, const uint8_t* pred, int size, int qp, int coeff_type, int fast, int n,
                                      int16_t* coefq, uint8_t* rec, int* cbp) {
  return kat_code_tu_batch<uint8_t>(org, pred, size, qp, coeff_type, fast, n, 8, coefq, rec, cbp);
}
extern "C" int thor_hip_code_tu_batch_hbd(const uint16_t* org, const uint16_t* pred, int size, int qp, int coeff_type, int fast, int n,
                                          int bitdepth, int16_t* coefq, uint16_t* rec, int* cbp) {
  if (bitdepth < 9 || bitdepth > 12) return 1;
  return kat_code_tu_batch<uint16_t>(org, pred, size, qp, coeff_type, fast, n, bitdepth, coefq, rec, cbp);
}

template <typename PIX> static int kat_deblock_frame(PIX* yuv, int width, int height, int qp, int bitdepth, const thor_hip_cell* cells) {
  static_assert(sizeof(thor_hip_cell) == sizeof(DbCell), "thor_hip_cell must mirror tk::DbCell");
  if (!yuv || !cells || width % 8 || height % 8 || width < 16 || height < 16 || qp < 0 || qp > 51) return 1;
  if (!ensure_init(g_inited ? g_device : 0)) return 3;
  DevFrame<PIX> f;
  f.alloc(width, height, 0);
  const size_t ncell = (size_t)(width / 4) * (height / 4);
  DbCell* d_cells = to_dev((const DbCell*)cells, ncell);
  const size_t B = sizeof(PIX);
  HIPCHECK(hipMemcpy2D(f.p.y, f.p.sy * B, yuv, width * B, width * B, height, hipMemcpyHostToDevice));
  PIX* hu = yuv + (size_t)width * height;
  PIX* hv = hu + (size_t)(width / 2) * (height / 2);
  HIPCHECK(hipMemcpy2D(f.p.u, f.p.sc * B, hu, width / 2 * B, width / 2 * B, height / 2, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy2D(f.p.v, f.p.sc * B, hv, width / 2 * B, width / 2 * B, height / 2, hipMemcpyHostToDevice));
  FrameJob<PIX> J;
  memset(&J, 0, sizeof(J));
  J.cfg.width = width; J.cfg.height = height; J.cfg.bitdepth = bitdepth;
  J.qp = qp; J.rec = f.p; J.cells = d_cells; J.cell_stride = width / 4;
  FrameJob<PIX>* d_job = to_dev(&J, 1);
  backend::run_deblock<PIX>(d_job, &J, 1);
  backend::dev_sync();
  HIPCHECK(hipMemcpy2D(yuv, width * B, f.p.y, f.p.sy * B, width * B, height, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy2D(hu, width / 2 * B, f.p.u, f.p.sc * B, width / 2 * B, height / 2, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy2D(hv, width / 2 * B, f.p.v, f.p.sc * B, width / 2 * B, height / 2, hipMemcpyDeviceToHost));
  backend::dev_free(d_job); backend::dev_free(d_cells);
  f.release();
  return 0;
}
extern "C" int thor_hip_deblock_frame(uint8_t* yuv, int width, int height, int qp, const thor_hip_cell* cells) {
  return kat_deblock_frame<uint8_t>(yuv, width, height, qp, 8, cells);
}
extern "C" int thor_hip_deblock_frame_hbd(uint16_t* yuv, int width, int height, int qp, int bitdepth, const thor_hip_cell* cells) {
  if (bitdepth < 9 || bitdepth > 12) return 1;
  return kat_deblock_frame<uint16_t>(yuv, width, height, qp, bitdepth, cells);
}

// ---- round 6: known-answer entry points for intra prediction, inter prediction of a whole block (luma + chroma, quadrant split), the
// bi-prediction average, chroma-from-luma, the CDEF direction search / filter, CLPF and the temporally interpolated reference --------------
namespace {
template <typename PIX>
int kat_intra(const PIX* plane, int width, int height, int stride, int bitdepth, int size, int tb_split, int n, const int* par, const PIX* rblocks, PIX* out) {
  if (!plane || !par || !out || n <= 0 || size < 4 || size > 64 || (size & (size - 1)) || (tb_split && !rblocks) || stride < width) return 1;
  const int cbs = tb_split ? 2 * size : size;
  for (int i = 0; i < n; i++) {
    const int* q = par + 7 * i;
    if (q[0] < 0 || q[1] < 0 || q[0] + cbs > height || q[1] + cbs > width || q[4] < 0 || q[5] < 0 || q[6] < 0 || q[5] + size > cbs || q[6] + size > cbs) return 2;
    if ((q[2] && q[1] + 2 * cbs > width) || (q[3] && q[0] + 2 * cbs > height)) return 2;   // up-right / down-left samples must exist
  }
  if (!ensure_init(g_inited ? g_device : 0)) return 3;
  PIX* d_p = to_dev(plane, (size_t)stride * height);
  int* d_par = to_dev(par, (size_t)7 * n);
  PIX* d_rb = tb_split ? to_dev(rblocks, (size_t)n * cbs * cbs) : nullptr;
  PIX* d_o = to_dev<PIX>(nullptr, (size_t)n * size * size);
  hipLaunchKernelGGL(k_kat_intra<PIX>, dim3(n), dim3(64), 0, g_stream, d_p, stride, bitdepth, size, tb_split, d_par, d_rb, d_o);
  HIPCHECK(hipGetLastError());
  backend::d2h(out, d_o, (size_t)n * size * size * sizeof(PIX));
  backend::dev_free(d_p); backend::dev_free(d_par); if (d_rb) backend::dev_free(d_rb); backend::dev_free(d_o);
  return 0;
}
// frame with the reference windows' replicate padding (what k_make_ref produces from a reconstruction)
template <typename PIX> DevFrame<PIX> kat_padded_ref(const PIX* yuv, int width, int height) {
  DevFrame<PIX> rec, ref;
  rec.alloc(width, height, 0);
  ref.alloc(width, height, kPadY);
  const size_t B = sizeof(PIX);
  const PIX* hu = yuv + (size_t)width * height;
  const PIX* hv = hu + (size_t)(width / 2) * (height / 2);
  HIPCHECK(hipMemcpy2D(rec.p.y, rec.p.sy * B, yuv, width * B, width * B, height, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy2D(rec.p.u, rec.p.sc * B, hu, width / 2 * B, width / 2 * B, height / 2, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy2D(rec.p.v, rec.p.sc * B, hv, width / 2 * B, width / 2 * B, height / 2, hipMemcpyHostToDevice));
  FrameJob<PIX> J;
  memset(&J, 0, sizeof(J));
  J.cfg.width = width; J.cfg.height = height; J.rec = rec.p;
  backend::run_make_ref<PIX>(&J, &ref.p, 1);
  backend::dev_sync();
  rec.release();
  return ref;
}
template <typename PIX>
int kat_inter_yuv(const PIX* yuv, int width, int height, int bitdepth, int size, int n, const int* par, const int16_t* mv, PIX* out) {
  if (!yuv || !par || !mv || !out || n <= 0 || size < 8 || size > 128 || (size & (size - 1)) || width % 8 || height % 8) return 1;
  for (int i = 0; i < n; i++) {
    const int* q = par + 5 * i;
    if (q[0] < 0 || q[1] < 0 || q[0] + size > height || q[1] + size > width || (q[4] && size < 16)) return 2;
  }
  if (!ensure_init(g_inited ? g_device : 0)) return 3;
  DevFrame<PIX> ref = kat_padded_ref(yuv, width, height);
  int* d_par = to_dev(par, (size_t)5 * n);
  int16_t* d_mv = to_dev(mv, (size_t)8 * n);
  const size_t per = (size_t)size * size * 3 / 2;
  PIX* d_o = to_dev<PIX>(nullptr, per * n);
  hipLaunchKernelGGL(k_kat_inter_yuv<PIX>, dim3(n), dim3(64), 0, g_stream, ref.p, width, height, bitdepth, size, d_par, d_mv, d_o);
  HIPCHECK(hipGetLastError());
  backend::d2h(out, d_o, per * n * sizeof(PIX));
  backend::dev_free(d_par); backend::dev_free(d_mv); backend::dev_free(d_o);
  ref.release();
  return 0;
}
template <typename PIX> int kat_average(const PIX* a, const PIX* b, int size, int n, PIX* out) {
  if (!a || !b || !out || n <= 0 || size < 8 || size > 128 || (size & (size - 1))) return 1;
  if (!ensure_init(g_inited ? g_device : 0)) return 3;
  const size_t tot = (size_t)n * size * size * 3 / 2;
  PIX* d_a = to_dev(a, tot);
  PIX* d_b = to_dev(b, tot);
  PIX* d_o = to_dev<PIX>(nullptr, tot);
  hipLaunchKernelGGL(k_kat_average<PIX>, dim3(n), dim3(64), 0, g_stream, d_a, d_b, size, d_o);
  HIPCHECK(hipGetLastError());
  backend::d2h(out, d_o, tot * sizeof(PIX));
  backend::dev_free(d_a); backend::dev_free(d_b); backend::dev_free(d_o);
  return 0;
}
template <typename PIX> int kat_cfl(const PIX* y, PIX* uv, const PIX* ry, int nl, int bitdepth, int n) {
  if (!y || !uv || !ry || n <= 0 || nl < 8 || nl > 128 || (nl & (nl - 1))) return 1;
  if (!ensure_init(g_inited ? g_device : 0)) return 3;
  const size_t ny = (size_t)n * nl * nl, nc = (size_t)n * 2 * (nl / 2) * (nl / 2);
  PIX* d_y = to_dev(y, ny);
  PIX* d_r = to_dev(ry, ny);
  PIX* d_uv = to_dev((const PIX*)uv, nc);
  hipLaunchKernelGGL(k_kat_cfl<PIX>, dim3(n), dim3(64), 0, g_stream, d_y, d_uv, d_r, nl, bitdepth);
  HIPCHECK(hipGetLastError());
  backend::d2h(uv, d_uv, nc * sizeof(PIX));
  backend::dev_free(d_y); backend::dev_free(d_r); backend::dev_free(d_uv);
  return 0;
}
template <typename PIX> int kat_cdef_dir(const PIX* blocks, int bitdepth, int n, int* dir, int* var) {
  if (!blocks || !dir || !var || n <= 0) return 1;
  if (!ensure_init(g_inited ? g_device : 0)) return 3;
  PIX* d_b = to_dev(blocks, (size_t)n * 64);
  int* d_d = to_dev<int>(nullptr, n);
  int* d_v = to_dev<int>(nullptr, n);
  hipLaunchKernelGGL(k_kat_cdef_dir<PIX>, dim3((n + 63) / 64), dim3(64), 0, g_stream, d_b, n, bitdepth - 8, d_d, d_v);
  HIPCHECK(hipGetLastError());
  backend::d2h(dir, d_d, (size_t)n * 4); backend::d2h(var, d_v, (size_t)n * 4);
  backend::dev_free(d_b); backend::dev_free(d_d); backend::dev_free(d_v);
  return 0;
}
template <typename PIX> int kat_cdef_filter(const PIX* plane, int w, int h, int stride, int bitdepth, int bsize, int n, const int* par, PIX* out) {
  if (!plane || !par || !out || n <= 0 || (bsize != 4 && bsize != 8) || stride < w) return 1;
  for (int i = 0; i < n; i++) {
    const int* q = par + 7 * i;
    if (q[0] < 0 || q[1] < 0 || q[0] + bsize > w || q[1] + bsize > h || q[4] < 0 || q[4] > 7) return 2;
  }
  if (!ensure_init(g_inited ? g_device : 0)) return 3;
  PIX* d_p = to_dev(plane, (size_t)stride * h);
  int* d_par = to_dev(par, (size_t)7 * n);
  PIX* d_o = to_dev<PIX>(nullptr, (size_t)n * bsize * bsize);
  hipLaunchKernelGGL(k_kat_cdef_filter<PIX>, dim3(n), dim3(64), 0, g_stream, d_p, w, h, stride, bsize, bitdepth - 8, d_par, d_o);
  HIPCHECK(hipGetLastError());
  backend::d2h(out, d_o, (size_t)n * bsize * bsize * sizeof(PIX));
  backend::dev_free(d_p); backend::dev_free(d_par); backend::dev_free(d_o);
  return 0;
}
template <typename PIX> void kat_upload(DevFrame<PIX>& f, const PIX* yuv, int width, int height) {
  const size_t B = sizeof(PIX);
  const PIX* hu = yuv + (size_t)width * height;
  const PIX* hv = hu + (size_t)(width / 2) * (height / 2);
  HIPCHECK(hipMemcpy2D(f.p.y, f.p.sy * B, yuv, width * B, width * B, height, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy2D(f.p.u, f.p.sc * B, hu, width / 2 * B, width / 2 * B, height / 2, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy2D(f.p.v, f.p.sc * B, hv, width / 2 * B, width / 2 * B, height / 2, hipMemcpyHostToDevice));
}
template <typename PIX> void kat_download(const DevFrame<PIX>& f, PIX* yuv, int width, int height) {
  const size_t B = sizeof(PIX);
  PIX* hu = yuv + (size_t)width * height;
  PIX* hv = hu + (size_t)(width / 2) * (height / 2);
  HIPCHECK(hipMemcpy2D(yuv, width * B, f.p.y, f.p.sy * B, width * B, height, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy2D(hu, width / 2 * B, f.p.u, f.p.sc * B, width / 2 * B, height / 2, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy2D(hv, width / 2 * B, f.p.v, f.p.sc * B, width / 2 * B, height / 2, hipMemcpyDeviceToHost));
}
// CLPF: the two device passes of the product (statistics per 8x8 block, filter per 8x8 luma / 4x4 chroma unit) on one frame.
template <typename PIX>
int kat_clpf(const PIX* rec_yuv, const PIX* org_yuv, int width, int height, int bitdepth, int qp, const thor_hip_cell* cells, const int* strength, int fb_log2,
             const uint8_t* fb_on, uint32_t* stats, PIX* out_yuv) {
  if (!rec_yuv || !org_yuv || !cells || !strength || !fb_on || !stats || !out_yuv || width % 16 || height % 16 || fb_log2 < 5 || fb_log2 > 7) return 1;
  if (!ensure_init(g_inited ? g_device : 0)) return 3;
  DevFrame<PIX> rec, src, org;
  rec.alloc(width, height, 0); src.alloc(width, height, 0); org.alloc(width, height, 0);
  kat_upload(rec, rec_yuv, width, height); kat_upload(src, rec_yuv, width, height); kat_upload(org, org_yuv, width, height);
  const size_t ncell = (size_t)(width / 4) * (height / 4);
  DbCell* d_cells = to_dev((const DbCell*)cells, ncell);
  const int nblk = (width / 8) * (height / 8) + 2 * (width / 16) * (height / 16);
  const int nfb = ((width + (1 << fb_log2) - 1) >> fb_log2) * ((height + (1 << fb_log2) - 1) >> fb_log2);
  uint32_t* d_stats = to_dev<uint32_t>(nullptr, (size_t)4 * nblk);
  uint8_t* d_on = to_dev(fb_on, (size_t)nfb);
  ClpfJob<PIX> J;
  memset(&J, 0, sizeof(J));
  J.rec = rec.p; J.src = src.p; J.org = org.p; J.width = width; J.height = height; J.bitdepth = bitdepth; J.qp = qp;
  J.cells = d_cells; J.cs = width / 4; J.stats = d_stats;
  for (int k = 0; k < 3; k++) J.strength[k] = strength[k];
  J.fb_log2 = fb_log2; J.fb_on = d_on;
  ClpfJob<PIX>* d_job = to_dev(&J, 1);
  backend::run_clpf_stats<PIX>(d_job, &J, 1);
  backend::run_clpf_apply<PIX>(d_job, &J, 1);
  backend::dev_sync();
  backend::d2h(stats, d_stats, (size_t)4 * nblk * 4);
  kat_download(rec, out_yuv, width, height);
  backend::dev_free(d_cells); backend::dev_free(d_stats); backend::dev_free(d_on); backend::dev_free(d_job);
  rec.release(); src.release(); org.release();
  return 0;
}
// interpolate_frames(new, ref0, ref1, 2, 1) (common/temporal_interp.c:909) through the engine's own path (Engine::make_interp_frames, tk_interp_dev.h)
template <typename PIX> int kat_interpolate(const PIX* yuv0, const PIX* yuv1, int width, int height, int bitdepth, PIX* out_yuv) {
  if (!yuv0 || !yuv1 || !out_yuv || width % 8 || height % 8 || width < 64 || height < 64) return 1;
  if (!ensure_init(g_inited ? g_device : 0)) return 3;
  SeqParams sp;
  sp.width = width; sp.height = height; sp.bitdepth = bitdepth; sp.input_bitdepth = bitdepth;
  sp.num_reorder_pics = 7; sp.interp_ref = 1; sp.max_num_ref = 2; sp.HQperiod = 8; sp.cdef = 0; sp.clpf = 0;
  Engine<PIX>* eng = new Engine<PIX>();
  eng->open(sp, 1);
  Stream<PIX>& q = eng->st[0];
  for (int k = 0; k < 2; k++) {
    kat_upload(q.rec, k ? yuv1 : yuv0, width, height);
    FrameJob<PIX> J;
    memset(&J, 0, sizeof(J));
    J.cfg.width = width; J.cfg.height = height; J.rec = q.rec.p;
    backend::run_make_ref<PIX>(&J, &q.ring[k].p, 1);
    backend::dev_sync();
  }
  std::vector<FrameParams> fp(1);
  fp[0].interp_ref = 1; fp[0].interp_src[0] = 0; fp[0].interp_src[1] = 1; fp[0].frame_num = 1;
  eng->make_interp_frames(fp, 0, 1);
  backend::dev_sync();
  kat_download(q.interp, out_yuv, width, height);
  eng->close();
  delete eng;
  return 0;
}
}  // namespace
#define KAT_BD(call8, call16) do { if (bitdepth == 8) return call8; if (bitdepth >= 9 && bitdepth <= 12) return call16; return 1; } while (0)
extern "C" int thor_hip_kat_intra(const void* plane, int width, int height, int stride, int bitdepth, int size, int tb_split, int n, const int* par,
                                  const void* rblocks, void* out) {
  KAT_BD(kat_intra<uint8_t>((const uint8_t*)plane, width, height, stride, 8, size, tb_split, n, par, (const uint8_t*)rblocks, (uint8_t*)out),
         kat_intra<uint16_t>((const uint16_t*)plane, width, height, stride, bitdepth, size, tb_split, n, par, (const uint16_t*)rblocks, (uint16_t*)out));
}
extern "C" int thor_hip_kat_inter_yuv(const void* yuv, int width, int height, int bitdepth, int size, int n, const int* par, const int16_t* mv, void* out) {
  KAT_BD(kat_inter_yuv<uint8_t>((const uint8_t*)yuv, width, height, 8, size, n, par, mv, (uint8_t*)out),
         kat_inter_yuv<uint16_t>((const uint16_t*)yuv, width, height, bitdepth, size, n, par, mv, (uint16_t*)out));
}
extern "C" int thor_hip_kat_average(const void* a, const void* b, int size, int bitdepth, int n, void* out) {
  KAT_BD(kat_average<uint8_t>((const uint8_t*)a, (const uint8_t*)b, size, n, (uint8_t*)out),
         kat_average<uint16_t>((const uint16_t*)a, (const uint16_t*)b, size, n, (uint16_t*)out));
}
extern "C" int thor_hip_kat_cfl(const void* y, void* uv, const void* ry, int n_luma, int bitdepth, int n) {
  KAT_BD(kat_cfl<uint8_t>((const uint8_t*)y, (uint8_t*)uv, (const uint8_t*)ry, n_luma, 8, n),
         kat_cfl<uint16_t>((const uint16_t*)y, (uint16_t*)uv, (const uint16_t*)ry, n_luma, bitdepth, n));
}
extern "C" int thor_hip_kat_cdef_dir(const void* blocks, int bitdepth, int n, int* dir, int* var) {
  KAT_BD(kat_cdef_dir<uint8_t>((const uint8_t*)blocks, 8, n, dir, var), kat_cdef_dir<uint16_t>((const uint16_t*)blocks, bitdepth, n, dir, var));
}
extern "C" int thor_hip_kat_cdef_filter(const void* plane, int width, int height, int stride, int bitdepth, int bsize, int n, const int* par, void* out) {
  KAT_BD(kat_cdef_filter<uint8_t>((const uint8_t*)plane, width, height, stride, 8, bsize, n, par, (uint8_t*)out),
         kat_cdef_filter<uint16_t>((const uint16_t*)plane, width, height, stride, bitdepth, bsize, n, par, (uint16_t*)out));
}
extern "C" int thor_hip_kat_clpf(const void* rec_yuv, const void* org_yuv, int width, int height, int bitdepth, int qp, const thor_hip_cell* cells,
                                 const int* strength, int fb_log2, const uint8_t* fb_on, uint32_t* stats, void* out_yuv) {
  KAT_BD(kat_clpf<uint8_t>((const uint8_t*)rec_yuv, (const uint8_t*)org_yuv, width, height, 8, qp, cells, strength, fb_log2, fb_on, stats, (uint8_t*)out_yuv),
         kat_clpf<uint16_t>((const uint16_t*)rec_yuv, (const uint16_t*)org_yuv, width, height, bitdepth, qp, cells, strength, fb_log2, fb_on, stats, (uint16_t*)out_yuv));
}
extern "C" int thor_hip_kat_interpolate(const void* yuv0, const void* yuv1, int width, int height, int bitdepth, void* out_yuv) {
  KAT_BD(kat_interpolate<uint8_t>((const uint8_t*)yuv0, (const uint8_t*)yuv1, width, height, 8, (uint8_t*)out_yuv),
         kat_interpolate<uint16_t>((const uint16_t*)yuv0, (const uint16_t*)yuv1, width, height, bitdepth, (uint16_t*)out_yuv));
}

// Resources of the superblock kernel as the runtime sees them (a guard against silent occupancy regressions: round 6 found the 8-bit kernel at 227 VGPRs =
// two workgroups per CU after an unrelated kernel had raised the register budget of a shared __noinline__ function).
extern "C" int thor_hip_superblock_kernel_info(int sample_bytes, int* num_regs, int* lds_bytes, int* private_bytes, int* workgroups_per_cu) {
  if (!ensure_init(g_inited ? g_device : 0)) return 3;
  hipFuncAttributes a;
  int per_cu = 0;
  if (sample_bytes == 1) {
    HIPCHECK(hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_superblocks<uint8_t>)));
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_superblocks<uint8_t>, kWgThreads, 0));
  } else if (sample_bytes == 2) {
    HIPCHECK(hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_superblocks<uint16_t>)));
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_superblocks<uint16_t>, kWgThreads, 0));
  } else if (sample_bytes == 0) {   // the latency build of the 8-bit kernel (thor_hip_lat.cpp)
    if (thor_lat_kernel_info(num_regs, lds_bytes, private_bytes)) return 2;
    if (workgroups_per_cu) *workgroups_per_cu = thor_lat_workgroups_per_cu();
    return 0;
  } else if (sample_bytes == 3) {   // the eight-wavefront build of the 8-bit kernel (thor_hip_wide.cpp)
    if (thor_wide_kernel_info(num_regs, lds_bytes, private_bytes)) return 2;
    if (workgroups_per_cu) *workgroups_per_cu = thor_wide_workgroups_per_cu();
    return 0;
  } else return 1;
  if (num_regs) *num_regs = a.numRegs;
  if (lds_bytes) *lds_bytes = (int)a.sharedSizeBytes;
  if (private_bytes) *private_bytes = (int)a.localSizeBytes;
  if (workgroups_per_cu) *workgroups_per_cu = per_cu;
  return 0;
}
// Which build of the 8-bit superblock kernel the engine configured last launches with: 0 throughput (thor_hip.cpp), 1 latency (thor_hip_lat.cpp), 2 eight
// wavefronts per workgroup (thor_hip_wide.cpp).  Decided at the engine's first launch from the number of streams and the geometry (run_superblocks).
extern "C" int thor_hip_superblock_kernel_in_use(void) { return tk::backend::g_last_kern; }
