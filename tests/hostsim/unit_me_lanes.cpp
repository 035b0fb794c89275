// unit_me_lanes.cpp - TEST INFRASTRUCTURE ONLY.  The motion search with one lane per candidate (tk_me.h: me_cand_fullpel / me_cand8_subpel, taken
// by teams of 64 lanes for 8-bit PUs of up to 32x32 samples) against the generic search (every other team size) on the CPU: the product's
// motion_estimate runs the same sequence of searches twice - once with a team of 64 lanes (64 OS threads, the cross-lane primitives go through the
// exchange below) and once with a 1-lane team - and vector and cost of every search must be equal.  On the MI355X the same comparison is made inside
// the kernel by -DTK_ME_CROSSCHECK builds (scripts/gpu_r5_call6.sh); this is its CPU twin, so that the lane-per-candidate code is not gated by GPU runs alone.
//   g++ -std=c++17 -O2 -fno-strict-aliasing -DTHOR_HOSTSIM -DTHOR_HOSTSIM_LANES=64 -ffp-contract=off -pthread -o unit_me_lanes tests/hostsim/unit_me_lanes.cpp
#include "../../thor_amd/csrc/tk_me.h"
#include "../../thor_amd/csrc/tk_tables.h"
#include <atomic>
#include <cmath>
#include <cstdio>
#include <thread>
#include <vector>
namespace tk {
Tables g_tab;
long long g_prune_stat[8];
namespace hostlanes {   // every lane is an OS thread; one exchange primitive (publish a value, read everybody's), as in tests/hostsim/hostsim.cpp
struct Shared { int n = 1; std::atomic<int> count{0}; std::atomic<int> sense{0}; unsigned long long slots[64]; };
static thread_local Shared* tl_sh = nullptr;
static thread_local int tl_rank = 0, tl_sense = 0;
int lanes() { return tl_sh ? tl_sh->n : 1; }
int rank() { return tl_rank; }
void barrier() {
  Shared* sh = tl_sh;
  if (!sh || sh->n == 1) return;
  const int my = tl_sense ^= 1;
  if (sh->count.fetch_add(1, std::memory_order_acq_rel) == sh->n - 1) { sh->count.store(0, std::memory_order_relaxed); sh->sense.store(my, std::memory_order_release); }
  else { int spins = 0; while (sh->sense.load(std::memory_order_acquire) != my) if (++spins > 200) { std::this_thread::yield(); spins = 0; } }
}
static thread_local unsigned long long tl_single[1];
const unsigned long long* exchange_begin(unsigned long long v) {
  Shared* sh = tl_sh;
  if (!sh || sh->n == 1) { tl_single[0] = v; return tl_single; }
  sh->slots[tl_rank] = v;
  barrier();
  return sh->slots;
}
void exchange_end() { barrier(); }
}  // namespace hostlanes
}  // namespace tk
using namespace tk;

enum { W = 320, H = 192, PITCH = W + 2 * kPadY };
struct Case { int pw, ph, cb; };
struct Result { unsigned cost; mv_t mv; };

// One sequence of searches (the per-"superblock" candidate list evolves along it) with a team of `lanes` lanes.
template <typename PIX>
static std::vector<Result> run_sequence(int lanes, const Case& c, int iters, const PIX* cur0, const PIX* ref0, int bipred, int sign, int window, unsigned seed, double lam = 9.5, int list_period = 8,
                                        int bitdepth = 8) {
  std::vector<Result> out(iters);
  static MeWs ws; static MeLists lists; static uint32_t win[2400]; alignas(16) static PIX org[32 * 32]; static long long prof[32];
  memset(&ws, 0, sizeof(ws)); memset(&lists, 0, sizeof(lists));
  ws.lists = &lists; ws.prof = prof; ws.win = window ? win : nullptr; ws.win_cap = window ? 4500 * (int)sizeof(PIX) : 0; ws.cwin_valid = 0;   // no window: every candidate is read from the plane
  hostlanes::Shared sh;
  sh.n = lanes;
  std::vector<std::thread> th;
  for (int r = 0; r < lanes; r++)
    th.emplace_back([&, r]() {
      hostlanes::tl_sh = &sh; hostlanes::tl_rank = r; hostlanes::tl_sense = 0;
      const Team t = mk_team(r, lanes);
      MeArgs a;
      a.cb_size = c.cb; a.ostride = c.pw; a.width = c.pw; a.height = c.ph; a.rstride = PITCH; a.sign = sign; a.fwidth = W; a.fheight = H;
      a.enable_bipred = bipred; a.bitdepth = bitdepth; a.speed = 0; a.lam = lam;
      unsigned rng = seed;
      for (int it = 0; it < iters; it++) {
        // positions all over the frame, its corners and edges included (clipped vectors, windows cut by the padding)
        rng = rng * 1664525u + 1013904223u;
        int bx = (int)((rng >> 8) % (unsigned)((W - c.pw) / 4 + 1)) * 4, by = (int)((rng >> 20) % (unsigned)((H - c.ph) / 4 + 1)) * 4;
        if (it % 7 == 0) { bx = it % 14 ? 0 : W - c.pw; by = it % 21 ? 0 : H - c.ph; }
        for (int k = r; k < c.pw * c.ph; k += lanes) org[k] = cur0[(by + k / c.pw) * PITCH + bx + k % c.pw];
        if (r == 0 && it % list_period == 0) { lists.mvcand_num[0] = 0; lists.mvcand_mask[0] = 0; }   // a new "superblock": empty candidate list
        t.sync();
        a.xpos = bx & ~(c.cb - 1); a.ypos = by & ~(c.cb - 1); a.pu_x = bx; a.pu_y = by;
        rng = rng * 1664525u + 1013904223u;
        // predictor / centre: the true motion, a little off, far off (outside the staged window), or zero
        mv_t mvp;
        switch (it & 3) {
          case 0: mvp = mk_mv(20, -12); break;
          case 1: mvp = mk_mv(8 + (int)((rng >> 9) & 7), -4 - (int)((rng >> 13) & 7)); break;
          case 2: mvp = mk_mv(0, 0); break;
          default: mvp = mk_mv((int)((rng >> 7) % 161) - 80, (int)((rng >> 17) % 97) - 48); break;
        }
        const mv_t mvc = (it % 5 == 4) ? mk_mv(mvp.x + 6, mvp.y - 9) : mvp;
        mv_t mv;
        const unsigned cost = (unsigned)motion_estimate<PIX, SP_LDS>(t, &ws, org, ref0 + by * PITCH + bx, a, mvc, mvp, 0, &mv);
        if (r == 0) { out[it].cost = cost; out[it].mv = mv; add_mvcand(&ws, 0, mv); }
        t.sync();
      }
    });
  for (auto& x : th) x.join();
  return out;
}

int main(int argc, char** argv) {
  // unit_me_lanes [FIRST_CASE [NUM_CASES]]: a slice of the PU shapes below (the whole list takes minutes: 64 OS threads per team)
  const int first = argc > 1 ? atoi(argv[1]) : 0, count = argc > 2 ? atoi(argv[2]) : 1000;
  init_tables(&g_tab);
  std::vector<uint8_t> cur((size_t)PITCH * (H + 2 * kPadY)), ref(cur.size());
  unsigned rng = 12345u;
  auto tex = [](int x, int y) -> int { return 128 + (int)(50.0 * sin(x * 0.07) * cos(y * 0.05) + 30.0 * sin(x * 0.31 + y * 0.23) + 14.0 * sin(x * 1.3) * sin(y * 1.1)); };
  auto sat = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
  // reference plane = texture inside the frame, replicated into the padding (as k_make_ref leaves it); current = displaced by (5, -3) samples + noise
  for (int y = 0; y < H + 2 * kPadY; y++)
    for (int x = 0; x < PITCH; x++) {
      const int fx = x - kPadY < 0 ? 0 : x - kPadY >= W ? W - 1 : x - kPadY, fy = y - kPadY < 0 ? 0 : y - kPadY >= H ? H - 1 : y - kPadY;
      rng = rng * 1664525u + 1013904223u;
      ref[(size_t)y * PITCH + x] = sat(tex(fx, fy));
      cur[(size_t)y * PITCH + x] = sat(tex(fx + 5, fy - 3) + (int)((rng >> 24) & 3) - 1);
    }
  const uint8_t* cur0 = cur.data() + (size_t)kPadY * PITCH + kPadY;
  const uint8_t* ref0 = ref.data() + (size_t)kPadY * PITCH + kPadY;
  const Case cases[] = {{4, 4, 8}, {8, 8, 8}, {8, 4, 8}, {4, 8, 8}, {16, 16, 16}, {16, 8, 16}, {8, 16, 16}, {32, 32, 32}, {32, 16, 32}, {16, 32, 32}, {32, 8, 32}, {8, 32, 32}};
  int fails = 0, total = 0;
  int ci = -1;
  for (const Case& c : cases) {
    ci++;
    if (ci < first || ci >= first + count) continue;
    // 0 plain / 1 reference "in the future" (vector signs) / 2 the other filter set (enable_bipred 0) / 3 no staged window /
    // 4 one long candidate list (never reset: up to 48 entries), large lambda / 5 small lambda, future reference, the other filter set
    for (int variant = 0; variant < 6; variant++) {
      const int iters = c.pw * c.ph >= 512 ? 24 : 48;
      const int sign = variant == 1 || variant == 5, bipred = variant != 2 && variant != 5, window = variant != 3;
      const double lam = variant == 4 ? 28.3 : variant == 5 ? 2.1 : 9.5;
      const int period = variant == 4 ? 1000 : 8;
      const unsigned seed = 777u + (unsigned)c.pw * 31u + (unsigned)c.ph + 1000u * (unsigned)variant;
      const std::vector<Result> a = run_sequence<uint8_t>(64, c, iters, cur0, ref0, bipred, sign, window, seed, lam, period);
      const std::vector<Result> b = run_sequence<uint8_t>(1, c, iters, cur0, ref0, bipred, sign, window, seed, lam, period);
      for (int it = 0; it < iters; it++) {
        total++;
        if (a[it].cost != b[it].cost || a[it].mv.x != b[it].mv.x || a[it].mv.y != b[it].mv.y) {
          if (fails++ < 20) fprintf(stderr, "FAIL PU %dx%d (CB %d) variant %d search %d: 64 lanes cost %u mv (%d, %d) != 1 lane cost %u mv (%d, %d)\n", c.pw, c.ph, c.cb, variant, it, a[it].cost,
                                    a[it].mv.x, a[it].mv.y, b[it].cost, b[it].mv.x, b[it].mv.y);
        }
      }
    }
  }
  // Round 6: the same comparison on 16-bit samples (bitdepth 10: me_cand_fullpel<uint16_t> - full-pel passes with one lane per candidate and v_sad_u16;
  // the sub-pel passes of 16-bit PUs stay generic), plain / future reference / no window / long list
  int total16 = 0;
  {
    std::vector<uint16_t> cur16(cur.size()), ref16(ref.size());
    unsigned r2 = 999u;
    for (size_t i = 0; i < cur.size(); i++) {
      r2 = r2 * 1664525u + 1013904223u;
      ref16[i] = (uint16_t)(ref[i] * 4 + ((r2 >> 20) & 3)); cur16[i] = (uint16_t)(cur[i] * 4 + ((r2 >> 24) & 3));
    }
    const uint16_t* c16 = cur16.data() + (size_t)kPadY * PITCH + kPadY;
    const uint16_t* r16 = ref16.data() + (size_t)kPadY * PITCH + kPadY;
    ci = -1;
    for (const Case& c : cases) {
      ci++;
      if (ci < first || ci >= first + count) continue;
      for (int variant : {0, 1, 3, 4}) {
        const int iters = c.pw * c.ph >= 512 ? 12 : 24;
        const int sign = variant == 1, window = variant != 3;
        const double lam = variant == 4 ? 28.3 : 9.5;
        const int period = variant == 4 ? 1000 : 8;
        const unsigned seed = 4242u + (unsigned)c.pw * 31u + (unsigned)c.ph + 1000u * (unsigned)variant;
        const std::vector<Result> a = run_sequence<uint16_t>(64, c, iters, c16, r16, 1, sign, window, seed, lam, period, 10);
        const std::vector<Result> b = run_sequence<uint16_t>(1, c, iters, c16, r16, 1, sign, window, seed, lam, period, 10);
        for (int it = 0; it < iters; it++) {
          total16++;
          if (a[it].cost != b[it].cost || a[it].mv.x != b[it].mv.x || a[it].mv.y != b[it].mv.y) {
            if (fails++ < 20) fprintf(stderr, "FAIL 16-bit PU %dx%d (CB %d) variant %d search %d: 64 lanes cost %u mv (%d, %d) != 1 lane cost %u mv (%d, %d)\n", c.pw, c.ph, c.cb, variant, it,
                                      a[it].cost, a[it].mv.x, a[it].mv.y, b[it].cost, b[it].mv.x, b[it].mv.y);
          }
        }
      }
    }
  }
  if (fails) { fprintf(stderr, "%d of %d searches differ\n", fails, total + total16); return 1; }
  printf("ok: %d searches (8-bit) + %d (16-bit), lane-per-candidate (64 lanes) == generic (1 lane)\n", total, total16);
  return 0;
}
