// tk_me.h - motion search for one prediction unit, team-cooperative.
// Specification followed: enc/encode_block.c:517-711 (motion_estimate, encoder_speed 0 path:
// telescope 32/16/8/4, per-SB candidate list with the 5-offset "widesad" for 16x16 CBs, hexagon
// refinement, 8 half-pel + 8 quarter-pel positions), :467-515 (quote_mv_bits), :417-453
// (sad_calc / widesad_calc), :69-82 (add_mvcandidate).  Every stage evaluates its candidates in
// parallel and then picks the winner by scanning costs in the reference's evaluation order with
// strict '<', which reproduces the sequential tie-breaking exactly.
#pragma once
#include "tk_common.h"
#include "tk_pred.h"
#include "tk_xform.h"

namespace tk {

enum { kMeWideChunk = 12, kMeMaxCand = kMeWideChunk * 5 };  // 5-offset SADs are evaluated 12 candidates at a time

// per-SB candidate lists (frame_info.mvcand[], enc/mainenc.h:146-148), reset per SB.  One instance per workgroup:
// the list of reference r is only ever touched by the wavefront that is searching reference r at that moment.
struct MeLists {
  mv_t mvcand[kMaxRefs][64];
  int mvcand_num[kMaxRefs];
  unsigned long long mvcand_mask[kMaxRefs];
  int best_ref;  // frame_info.best_ref (enc/mainenc.h:143): per-SB state of the encoder_speed 2 reference shortcut
};
struct MeWs {  // per wavefront
  int sad[kMeMaxCand];
  mv_t cmv[64];
  MeLists* lists;
  long long* prof;
  uint32_t* win;  // per-wave LDS for the search window (see MeWin below), nullptr: none
  int win_cap;    // its size in bytes
  // A window staged for a whole coding block (me_stage_cb_window): the searches of a reference's HOR / VER / QUAD partitions all start from
  // the same centre, so one (CB + 2R)^2 window serves all eight of them.  cwin_ax / cwin_ay: absolute luma position of its first sample.
  int cwin_valid, cwin_ref, cwin_ax, cwin_ay, cwin_Ww, cwin_Wh, cwin_pitch;
};
TK_DEV int mv_len1(int a) {
  // (selects on values computed up front: as early returns this compiled to four nested exec-masked branches per vector component and candidate)
  a = iabs(a);
  const int big = 10 + ((a - 36) >> 4) + 1;   // a >= 36
  const int mid = 5 + ((a - 4) >> 3) + 1;     // 4 <= a < 36
  int r = a < 36 ? mid : big;
  r = a < 4 ? 5 : r;
  r = a < 2 ? 4 : r;
  r = a < 1 ? 2 : r;
  return r;
}
TK_DEV int quote_mv_bits(int dy, int dx) { return mv_len1(dx) + mv_len1(dy); }
TK_DEV unsigned mv_cost(double lam, int dy, int dx) {
  return (unsigned)mul_add_nofma(lam, (double)quote_mv_bits(dy, dx), 0.5);
}
// add_mvcandidate (encode_block.c:69-82) - call from ONE lane.
TK_DEV void add_mvcand(MeWs* w_, int r, mv_t mv) {
  const auto w = ldsc(lds_ld(&w_->lists));
  mv_t imv = mk_mv((mv.x + 2) >> 2, (mv.y + 2) >> 2);
  unsigned long long m = 1ull << ((((int)imv.y << 3) ^ (int)imv.x) & 63);
  if (!(m & w->mvcand_mask[r])) {
    const int n = w->mvcand_num[r];
    w->mvcand[r][n].x = imv.x; w->mvcand[r][n].y = imv.y;
    w->mvcand_num[r] = n + 1;
  }
  w->mvcand_mask[r] |= m;
}

// Evaluate n candidates and return min over (cost << 32 | index): the first candidate in evaluation
// order among those with the smallest cost - exactly the winner of the reference's sequential
// "if (cost < min) ..." scan.  prep(c) -> per-candidate context, item(ctx, r) -> partial SAD of work
// item r < nit, cost(c, ctx, sad) -> cost.  G lanes share a candidate, partial sums are combined with
// xor-shuffles, the final minimum with a 64-bit wave reduction; no LDS traffic, no barriers.
template <class PrepF, class ItemF, class CostF>
TK_DEV unsigned long long eval_min(const Team t, int n, int nit, PrepF prep, ItemF item, CostF cost) {
  const int G = nit < t.size ? nit : t.size;
  const int P = t.size / G;
  const int slot = t.rank / G, sub = t.rank - slot * G;
  unsigned long long best = ~0ull;
  for (int c0 = 0; c0 < n; c0 += 2 * P) {
    const int ca = c0 + slot, cb = c0 + P + slot;
    const int va = ca < n, vb = cb < n;
    auto xa = prep(va ? ca : 0);
    auto xb = prep(vb ? cb : 0);
    int la = 0, lb = 0;
    if (va) for (int r = sub; r < nit; r += G) la += item(xa, r);
    if (vb) for (int r = sub; r < nit; r += G) lb += item(xb, r);
    la = team_group_sum(t, la, G); lb = team_group_sum(t, lb, G);
    if (sub == 0) {
      if (va) { unsigned long long k = ((unsigned long long)cost(ca, xa, la) << 32) | (unsigned)ca; best = k < best ? k : best; }
      if (vb) { unsigned long long k = ((unsigned long long)cost(cb, xb, lb) << 32) | (unsigned)cb; best = k < best ? k : best; }
    }
  }
  return TKU64(team_min64(t, best));
}

// Row segment of a block: up to 16 bytes (16 8-bit / 8 16-bit samples) held in four dwords, unused dwords zero.
struct Seg16 { uint32_t d[4]; };
#if !TK_HOST
typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
typedef uint32_t __attribute__((ext_vector_type(2))) u32x2;
typedef u32x4 __attribute__((aligned(1), may_alias)) u32x4_unaligned;
typedef u32x2 __attribute__((aligned(1), may_alias)) u32x2_unaligned;
#endif
// NB (4, 8 or 16) bytes at p.  SP: address space of p; LDS / scratch blocks are aligned to the segment size, frame planes
// (global) may be read at any byte offset.
template <int SP, int NB> TK_DEV Seg16 seg_load(const void* p) {
  Seg16 r;
  r.d[0] = r.d[1] = r.d[2] = r.d[3] = 0;
#if TK_HOST
  __builtin_memcpy(&r, p, (size_t)NB);
#else
  if constexpr (SP == SP_LDS) {
    const auto q = (const TK_LDS uint8_t*)(uint32_t)(uintptr_t)p;
    if constexpr (NB == 16) { const u32x4 v = *(const TK_LDS u32x4*)q; r.d[0] = v.x; r.d[1] = v.y; r.d[2] = v.z; r.d[3] = v.w; }
    else if constexpr (NB == 8) { const u32x2 v = *(const TK_LDS u32x2*)q; r.d[0] = v.x; r.d[1] = v.y; }
    else r.d[0] = *(const TK_LDS uint32_t*)q;
  } else {
    const auto q = (const TK_GLOBAL uint8_t*)p;
    if constexpr (NB == 16) { const u32x4 v = *(const TK_GLOBAL u32x4_unaligned*)q; r.d[0] = v.x; r.d[1] = v.y; r.d[2] = v.z; r.d[3] = v.w; }
    else if constexpr (NB == 8) { const u32x2 v = *(const TK_GLOBAL u32x2_unaligned*)q; r.d[0] = v.x; r.d[1] = v.y; }
    else r.d[0] = *(const TK_GLOBAL u32_unaligned*)q;
  }
#endif
  return r;
}
// sum of absolute sample differences of two NB-byte segments, added to acc
template <typename PIX, int NB> TK_DEV int seg_sad(const Seg16& a, const Seg16& b, int acc) {
#if TK_HOST
  const PIX* x = (const PIX*)a.d;
  const PIX* y = (const PIX*)b.d;
  for (int k = 0; k < (int)(NB / sizeof(PIX)); k++) acc += iabs((int)x[k] - (int)y[k]);
  return acc;
#else
  unsigned s = (unsigned)acc;
  if constexpr (sizeof(PIX) == 1) { for (int k = 0; k < NB / 4; k++) s = __builtin_amdgcn_sad_u8(a.d[k], b.d[k], s); }    // 4 samples per lane-op
  else { for (int k = 0; k < NB / 4; k++) s = __builtin_amdgcn_sad_u16(a.d[k], b.d[k], s); }                             // 2 samples per lane-op
  return (int)s;
#endif
}

// Truncating average (a + b) >> 1 per sample of two segments (bi-prediction, inter_prediction.c:228-247): per dword
// (a & b) + (((a ^ b) >> 1) & M), M = every bit but each sample's top one.
template <typename PIX> TK_DEV Seg16 seg_avg(const Seg16& a, const Seg16& b) {
  const uint32_t M = sizeof(PIX) == 1 ? 0x7f7f7f7fu : 0x7fff7fffu;
  Seg16 r;
  for (int k = 0; k < 4; k++) r.d[k] = (a.d[k] & b.d[k]) + (((a.d[k] ^ b.d[k]) >> 1) & M);
  return r;
}
// A candidate type with a second reference pointer `p2` is bi-predicted: its block is the truncating average of the blocks at p and p2
// (same stride); such candidates always read the planes.
template <class T, class = void> struct CandHasP2 { enum { value = 0 }; };
template <class T> struct CandHasP2<T, decltype((void)((T*)nullptr)->p2)> { enum { value = 1 }; };

// LDS search window of one motion search: the (w + 2R) x (h + 2R) samples of the reference plane around the search centre, staged
// once per search with coalesced 16-byte row loads; the telescope, candidate-list, 5-offset, hexagon and sub-pel passes whose
// blocks lie inside read it with aligned ds_read + v_alignbyte instead of gathering from the vector L1 (one coalesced global
// round trip per search instead of one gather round trip per pass; profiles/r03_ubench_l1gather.log).  Samples of 1 or 2 bytes;
// the reach R is the largest multiple of 4 up to kMeWinR for which the window fits the wave's LDS budget (MeWs::win_cap), at
// least kMeWinRmin - otherwise the search reads the plane.  Row pitch = row bytes + 4: consecutive rows start in different banks.
// Origin (ox, oy) is relative to the PU's co-located position in the reference plane.  The window lives in the wave's transform
// workspace (idle during a search) and the bytes that follow it (SmallWs::win_extra).
struct MeWin {
  const uint32_t* w32;
  int ox, oy, Ww, Wh;   // samples
  int pitch;            // bytes
  int on;
};
enum { kMeWinR = 20, kMeWinRmin = 8 };
TK_DEV int me_win_bytes(int w, int h, int R, int S) { return ((w + 2 * R) * S + 4) * (h + 2 * R) + 4; }
// NB bytes at byte offset `off` of the window (any alignment): NB/4 + 1 aligned dwords, funnel-shifted
template <int NB> TK_DEV Seg16 win_seg(const uint32_t* w32, int off) {
  Seg16 r;
  r.d[0] = r.d[1] = r.d[2] = r.d[3] = 0;
  const int d = off >> 2;
  const unsigned sh = (unsigned)(off & 3);
  uint32_t a[NB / 4 + 1];
#if TK_HOST
  for (int k = 0; k <= NB / 4; k++) a[k] = w32[d + k];
  for (int k = 0; k < NB / 4; k++) r.d[k] = (uint32_t)((((unsigned long long)a[k + 1] << 32) | a[k]) >> (8 * sh));
#else
  const TK_LDS uint32_t* l = (const TK_LDS uint32_t*)(uint32_t)(uintptr_t)w32 + d;
#pragma unroll
  for (int k = 0; k <= NB / 4; k++) a[k] = l[k];
#pragma unroll
  for (int k = 0; k < NB / 4; k++) r.d[k] = __builtin_amdgcn_alignbyte(a[k + 1], a[k], sh);
#endif
  return r;
}

// Core of the full-pel passes: SAD of the org block against n candidate blocks; sink(c, x, sad, mine) is called in every lane
// for every evaluated candidate slot (mine = this lane reports candidate c: first lane of its group, c < n).
// Work item = one row segment of a candidate block (up to 16 bytes: ONE memory instruction per lane instead of one per four
// samples).  PUs of up to `team size` segments (8-bit: everything up to 32x32): one segment per lane and candidate,
// G = segments-per-candidate lanes form a group, team/G candidates are evaluated side by side and up to four such candidate
// sets are in flight per lane; the group sum is a DPP butterfly.  Larger PUs: the whole team works on one candidate, four
// segments per lane in flight.  cand(c) -> {clipped mv, displacement (dx, dy), pointer to the displaced reference block}.
// An iteration whose candidate blocks all lie inside the staged window reads LDS, otherwise the reference plane.
// One iteration of the small-PU path: U candidate sets (U * P candidates) starting at candidate c0.  Straight-line code: the
// U reference segments are fetched back to back (window or plane, decided once for all of them) before the first SAD; slots
// beyond n evaluate candidate 0 and are masked out in the sink.
template <int SP, typename PIX, int NB, int U, class CandF, class SinkF>
TK_DEV void seg_sads_iter(const Team t, int n, int c0, int P, int G, int slot, int sub, const Seg16& o, int roff, int woff, int width, int height,
                          const MeWin& win, CandF cand, SinkF sink) {
  Seg16 r[U];
  decltype(cand(0)) x[U];
  int outside = 0;
#if !TK_HOST
#pragma unroll
#endif
  for (int u = 0; u < U; u++) {
    const int c = c0 + u * P + slot;
    x[u] = cand(c < n ? c : 0);
    outside |= !(x[u].dx >= win.ox && x[u].dx + width <= win.ox + win.Ww && x[u].dy >= win.oy && x[u].dy + height <= win.oy + win.Wh);
  }
  enum { BI = CandHasP2<decltype(cand(0))>::value };
  const int use_win = !BI && win.on && team_ballot(t, outside) == 0ull;
  if (use_win) {
#if !TK_HOST
#pragma unroll
#endif
    for (int u = 0; u < U; u++) r[u] = win_seg<NB>(win.w32, mul24(x[u].dy, win.pitch) + x[u].dx * (int)sizeof(PIX) + woff);
  } else {
#if !TK_HOST
#pragma unroll
#endif
    for (int u = 0; u < U; u++) r[u] = seg_load<SP_GLOBAL, NB>(x[u].p + roff);
    if constexpr (BI) {
#if !TK_HOST
#pragma unroll
#endif
      for (int u = 0; u < U; u++) r[u] = seg_avg<PIX>(r[u], seg_load<SP_GLOBAL, NB>(x[u].p2 + roff));
    }
  }
#if !TK_HOST
#pragma unroll
#endif
  for (int u = 0; u < U; u++) {
    const int c = c0 + u * P + slot;
    const int sad = team_group_sum(t, seg_sad<PIX, NB>(o, r[u], 0), G);
    sink(c, x[u], sad, c < n && sub == 0);
  }
}
template <int SP, typename PIX, int NB, class CandF, class SinkF>
TK_DEV void seg_sads_nb(const Team t, int n_, const PIX* org, int ostride, int rstride, int width, int height, const MeWin& win,
                        CandF cand, SinkF sink) {
  // wave-uniform scalars (function arguments arrive in vector registers: without this every branch below is exec-mask code)
  const int n = TKU(n_), tsz = TKU(t.size);
  const int lw = NB / (int)sizeof(PIX);              // samples per segment
  const int lgr = TKU(ilog2((unsigned)(width / lw)));   // log2(segments per row)
  const int nit = height << lgr;                       // segments per candidate
  const int G = nit < tsz ? nit : tsz;                 // powers of two
  const int lgG = TKU(ilog2((unsigned)G));
  const int P = tsz >> lgG;
  const int slot = t.rank >> lgG, sub = t.rank & (G - 1);
  if (nit <= tsz) {
    const int i = sub >> lgr, j = (sub & ((1 << lgr) - 1)) * lw;
    // (24-bit multiplies throughout the passes: row / pitch products are small, and v_mul_lo_u32 runs at a quarter of the rate)
    const Seg16 o = seg_load<SP, NB>(org + mul24(i, ostride) + j);
    const int roff = mul24(i, rstride) + j;
    const int woff = mul24(i - win.oy, win.pitch) + (j - win.ox) * (int)sizeof(PIX);   // bytes
    if (n <= P) seg_sads_iter<SP, PIX, NB, 1>(t, n, 0, P, G, slot, sub, o, roff, woff, width, height, win, cand, sink);
    else if (n <= 2 * P) seg_sads_iter<SP, PIX, NB, 2>(t, n, 0, P, G, slot, sub, o, roff, woff, width, height, win, cand, sink);
    else
      for (int c0 = 0; c0 < n; c0 += 4 * P) seg_sads_iter<SP, PIX, NB, 4>(t, n, c0, P, G, slot, sub, o, roff, woff, width, height, win, cand, sink);
  } else {
    const int ipl = nit >> lgG;  // a multiple of 4 except on teams smaller than a wavefront (host simulation)
    for (int c = 0; c < n; c++) {
      const auto x = cand(c);
      // the whole wave works on this candidate: one wave-uniform decision whether its block lies inside the staged window
      const int use_win = TKU(!CandHasP2<decltype(cand(0))>::value && win.on && x.dx >= win.ox && x.dx + width <= win.ox + win.Ww && x.dy >= win.oy && x.dy + height <= win.oy + win.Wh);
      const int wbase = mul24(x.dy - win.oy, win.pitch) + (x.dx - win.ox) * (int)sizeof(PIX);
      int sad = 0;
      for (int k0 = 0; k0 < ipl; k0 += 4) {
        Seg16 o[4], r[4];
#if !TK_HOST
#pragma unroll
#endif
        for (int k = 0; k < 4; k++)
          if (k0 + k < ipl) {
            const int q = sub + (k0 + k) * G, i = q >> lgr, j = (q & ((1 << lgr) - 1)) * lw;
            o[k] = seg_load<SP, NB>(org + mul24(i, ostride) + j);
            if (use_win) r[k] = win_seg<NB>(win.w32, wbase + mul24(i, win.pitch) + j * (int)sizeof(PIX));
            else r[k] = seg_load<SP_GLOBAL, NB>(x.p + mul24(i, rstride) + j);
            if constexpr (CandHasP2<decltype(cand(0))>::value) r[k] = seg_avg<PIX>(r[k], seg_load<SP_GLOBAL, NB>(x.p2 + mul24(i, rstride) + j));
          }
#if !TK_HOST
#pragma unroll
#endif
        for (int k = 0; k < 4; k++)
          if (k0 + k < ipl) sad = seg_sad<PIX, NB>(o[k], r[k], sad);
      }
      sad = team_group_sum(t, sad, G);
      sink(c, x, sad, sub == 0);
    }
  }
}
template <int SP, typename PIX, class CandF, class SinkF>
TK_DEV void seg_sads(const Team t, int n, const PIX* org, int ostride, int rstride, int width, int height, const MeWin& win, CandF cand, SinkF sink) {
  const int nb = (width < 16 / (int)sizeof(PIX) ? width : 16 / (int)sizeof(PIX)) * (int)sizeof(PIX);  // bytes per row segment
  if (nb == 16) seg_sads_nb<SP, PIX, 16>(t, n, org, ostride, rstride, width, height, win, cand, sink);
  else if (nb == 8) seg_sads_nb<SP, PIX, 8>(t, n, org, ostride, rstride, width, height, win, cand, sink);
  else seg_sads_nb<SP, PIX, 4>(t, n, org, ostride, rstride, width, height, win, cand, sink);
}
// Full-pel candidate evaluation: min over the n candidates of (cost << 32 | index) - the first candidate in evaluation order
// among those with the smallest cost, i.e. the winner of the reference's sequential strict-'<' scan.
template <int SP, typename PIX, class CandF, class CostF>
TK_DEV unsigned long long eval_fullpel(const Team t, int n, const PIX* org, int ostride, int rstride, int width, int height,
                                       const MeWin& win, CandF cand, CostF cost) {
  // Costs fit 24 bits - the SAD is at most 128 * 128 * 255 after the bit-depth shift, the vector cost at most sqrt(lambda) * 2 * mv_len1(65535) <
  // 120 * 8208 - and n <= 64: the minimum over (cost << 8 | index) is ONE 32-bit wave reduction (4 DPP v_min + 4 v_readlane) instead of a 64-bit
  // one; the host simulation asserts the bound.
  unsigned best32 = ~0u;
  seg_sads<SP>(t, n, org, ostride, rstride, width, height, win, cand, [&](int c, const decltype(cand(0))& x, int sad, int mine) {
    const unsigned cst = cost(x, sad);
#if TK_HOST
    if (mine && ((cst >> 24) != 0u || c > 255)) { fprintf(stderr, "eval_fullpel: cost %u / index %d does not fit the packed key\n", cst, c); abort(); }
#endif
    unsigned k32 = (cst << 8) | (unsigned)c;
    if (!mine) k32 = ~0u;
    best32 = k32 < best32 ? k32 : best32;
  });
  const unsigned m = team_min32(t, best32);
  return m == ~0u ? ~0ull : (((unsigned long long)(m >> 8)) << 32) | (m & 0xffu);
}

struct MeArgs {
  int cb_size;           // `size` argument of motion_estimate = CB size
  int ostride;           // stride of the original-sample block
  int width, height;     // PU dims
  int rstride;
  int sign, fwidth, fheight, xpos, ypos;  // CB position (Appendix B.16)
  int pu_x, pu_y;        // PU position (absolute, luma samples): only the LDS search window needs it
  int enable_bipred, bitdepth;
  int speed;             // encoder_speed (0 slow .. 2 fast)
  double lam;            // sqrt(lambda)
};


// ---------------------------------------------------------------------------------------------------------------------------------
// Full-pel search of an 8-bit PU of up to 32x32 samples with ONE LANE PER CANDIDATE (round 5).
// tools/ubench_me.cpp: with nothing else on the CU a search of a 4x4 PU costs the generic passes of motion_estimate 21 k cycles (4.3 k
// per telescope step, 22 k per step for 32x32).  A pass there is ~450 wave-instructions at ~5 cycles each, and most of them are not sample
// work: the lanes of a candidate GROUP (one row segment per lane) all form the candidate's vector, clip it, price it (two vector-bit
// counts + a double-precision multiply-add: ~45 instructions) - and a lane does that for every candidate SET of the pass (4 per lane for
// an 8x8 PU, 24 for a 32x32 one).  A wavefront issues one vector instruction per 4 clocks whatever the lanes do, so the instruction
// count per lane is the time.  Here lane c IS candidate c of the pass (25 grid points, <= 64 list entries, 6 hexagon points): it forms,
// clips and prices its vector ONCE and walks the rows of the block itself - per 16-byte row segment one broadcast read of the original
// (same address in every lane), one unaligned read of its own displaced segment from the staged window (or the plane) and v_sad_u8.
// 8x8: ~130 instructions per pass instead of ~450; 32x32: ~900 instead of ~5 000.  No cross-lane work except the final minimum.
// Same passes, same order, same costs, winner = min over (cost, evaluation index) = the reference's sequential strict-'<' scan
// (enc/encode_block.c:517-616); no duplicate-candidate bookkeeping (a vector evaluated twice cannot win twice).
//   NB: bytes per row segment (4, 8: the PU width; 16: widths 16 and 32 = one or two segments per row)
//   Round 6: PIX = uint16_t too (the reference's _hbd searches: SAD >> (bitdepth - 8), enc/encode_block.c:417-428): the same walk with v_sad_u16 on
//   16-byte segments of eight samples - one (8 wide), two (16) or four (32) segments per row, `sh` = bitdepth - 8.
template <typename PIX, int NB, int SP>
TK_DEVNI unsigned long long me_cand_fullpel(const Team t, MeWs* w_, const PIX* org_, const PIX* ref, int a_cb, int a_ostride, int a_width, int a_height,
                                            int a_rstride, int a_sign, int a_fw, int a_fh, int a_xpos, int a_ypos, double a_lam, const uint32_t* win_w32, int win_ox,
                                            int win_oy, int win_Ww, int win_Wh, int win_pitch, int win_on, mv_t mvc, mv_t mvp, int ref_idx, int a_sh) {
  constexpr int S = (int)sizeof(PIX), SPS = 16 / S;   // bytes per sample, samples per 16-byte segment
  const int sh = tk_uniform(a_sh);
  // (scalars one by one and the result in registers: a struct - by reference or by value - is a trip through the caller's stack in scratch memory)
  struct { int cb_size, ostride, width, height, rstride, sign, fwidth, fheight, xpos, ypos; double lam; } a_in = {a_cb, a_ostride, a_width, a_height, a_rstride, a_sign, a_fw, a_fh, a_xpos, a_ypos, a_lam};
  MeWin win_in;
  win_in.w32 = win_w32; win_in.ox = win_ox; win_in.oy = win_oy; win_in.Ww = win_Ww; win_in.Wh = win_Wh; win_in.pitch = win_pitch; win_in.on = win_on;
  const auto lists = ldsc(lds_ld(&w_->lists));
  // wave-uniform scalars
  const int cb = tk_uniform(a_in.cb_size), ostride = tk_uniform(a_in.ostride), width = tk_uniform(a_in.width), height = tk_uniform(a_in.height);
  const int rstride = tk_uniform(a_in.rstride), sign = tk_uniform(a_in.sign), fw = tk_uniform(a_in.fwidth), fh = tk_uniform(a_in.fheight);
  const int xpos = tk_uniform(a_in.xpos), ypos = tk_uniform(a_in.ypos);
  const double lam = tk_uniform_f64(a_in.lam);
  MeWin win;
  win.w32 = tk_uniform_ptr(win_in.w32); win.ox = tk_uniform(win_in.ox); win.oy = tk_uniform(win_in.oy); win.Ww = tk_uniform(win_in.Ww);
  win.Wh = tk_uniform(win_in.Wh); win.pitch = tk_uniform(win_in.pitch); win.on = tk_uniform(win_in.on);
  mvc = mk_mv(tk_uniform(mvc.x), tk_uniform(mvc.y));
  mvp = mk_mv(tk_uniform(mvp.x), tk_uniform(mvp.y));
  ref_idx = tk_uniform(ref_idx);
  org_ = tk_uniform_ptr(org_);
  ref = tk_uniform_ptr(ref);
  const int s = sign ? -1 : 1;
  const int spr = NB == 16 ? ((width * S) >> 4) : 1;   // 16-byte segments per row
  unsigned min_sad = kCostInit;
  mv_t mv_opt = mk_mv(0, 0);
  mv_t mv_ref = mk_mv(((mvc.x + 2) >> 2) << 2, ((mvc.y + 2) >> 2) << 2);
  auto clip_free = [&](mv_t ctr, int R) -> int {   // motion_estimate's test: no vector within +-R quarter-pels of ctr needs clipping
    const int ext = kPadY - 16, cy = s * ctr.y, cx = s * ctr.x;
    return ypos + ((cy - R) >> 2) >= -ext && ypos + ((cy + R + 3) >> 2) + cb <= fh + ext && xpos + ((cx - R) >> 2) >= -ext && xpos + ((cx + R + 3) >> 2) + cb <= fw + ext;
  };
  // SAD of the block displaced by (dx + off, dy) against the original: the lane's own walk over the rows, four rows in flight (heights are
  // multiples of four); window / plane and one / two segments per row are decided outside the loop (straight-line bodies: all eight or
  // sixteen reads of an iteration are issued before the first SAD waits for them)
  auto rows_sad = [&](auto win_tag, auto spr_tag, int dx, int dy, int off) -> unsigned {
    constexpr int WIN = decltype(win_tag)::value, SPR = decltype(spr_tag)::value, ROWS = SPR >= 4 ? 1 : 4 / SPR;   // four segments in flight
    unsigned sad = 0;
    int wb = mul24(dy - win.oy, win.pitch) + (dx + off - win.ox) * S;   // bytes
    const PIX* gb = ref + mul24(dy, rstride) + (dx + off);
    const PIX* ob = org_;
    for (int i = 0; i < height; i += ROWS) {
      Seg16 o[ROWS * SPR], r[ROWS * SPR];
#if !TK_HOST
#pragma unroll
#endif
      for (int k = 0; k < ROWS; k++)
#if !TK_HOST
#pragma unroll
#endif
        for (int sg = 0; sg < SPR; sg++) {
          o[k * SPR + sg] = seg_load<SP, NB>(ob + mul24(k, ostride) + SPS * sg);   // the same address in every lane
          if constexpr (WIN) r[k * SPR + sg] = win_seg<NB>(win.w32, wb + mul24(k, win.pitch) + 16 * sg);
          else r[k * SPR + sg] = seg_load<SP_GLOBAL, NB>(gb + mul24(k, rstride) + SPS * sg);
        }
#if !TK_HOST
#pragma unroll
#endif
      for (int q = 0; q < ROWS * SPR; q++) sad = (unsigned)seg_sad<PIX, NB>(o[q], r[q], (int)sad);
      wb += ROWS * win.pitch; gb += ROWS * rstride; ob += ROWS * ostride;
    }
    return sad;
  };
  struct T0 { enum { value = 0 }; };
  struct T1 { enum { value = 1 }; };
  struct T2 { enum { value = 2 }; };
  struct T4 { enum { value = 4 }; };
  auto block_sad = [&](int use_win, int dx, int dy, int off) -> unsigned {
    if constexpr (NB == 16) {
      if (spr == 2) return use_win ? rows_sad(T1(), T2(), dx, dy, off) : rows_sad(T0(), T2(), dx, dy, off);
      if constexpr (S == 2) { if (spr == 4) return use_win ? rows_sad(T1(), T4(), dx, dy, off) : rows_sad(T0(), T4(), dx, dy, off); }
    }
    return use_win ? rows_sad(T1(), T1(), dx, dy, off) : rows_sad(T0(), T1(), dx, dy, off);
  };
  // Cost of THIS LANE's candidate (vector m, not yet clipped; `valid` lanes only - the others return ~0u).  Every vector is clipped with clip_mv,
  // which leaves a vector inside the clip-free area alone: the same vectors motion_estimate evaluates with or without its `noclip` short cut.
  // WIDE (16x16 coding blocks, candidate list): the cost is that of the best x offset of {-3, -1, 0, 1, 3} (first minimum) with the vector moved
  // there (encode_block.c:430-453); *osel = that offset's index.
  auto lane_cost = [&](mv_t m, int valid, auto wide_tag, unsigned* osel) -> unsigned {
    constexpr int WIDE = decltype(wide_tag)::value;
    m = clip_mv(m, ypos, xpos, fw, fh, cb, cb, sign);
    {  // lanes without a candidate evaluate lane 0's vector (always a candidate) and drop the result: their own may point outside the staged window
      const int mp0 = team_bcast0(t, (int)(uint16_t)m.x | ((int)m.y << 16));
      if (!valid) m = mk_mv((int16_t)(mp0 & 0xffff), mp0 >> 16);
    }
    const int dx = s * (m.x >> 2), dy = s * (m.y >> 2);
    const int x0 = dx - (WIDE ? 3 : 0), x1 = dx + (WIDE ? 3 : 0);
    const int outside = valid && !(x0 >= win.ox && x1 + width <= win.ox + win.Ww && dy >= win.oy && dy + height <= win.oy + win.Wh);
    const int use_win = win.on && team_ballot(t, outside) == 0ull;
    unsigned sad;
    int mx = m.x;
    if constexpr (WIDE) {
      sad = 1u << 31;
      int bx = 0;
      for (int q = 0; q < 5; q++) {
        const int off = q == 0 ? -3 : q == 1 ? -1 : q == 2 ? 0 : q == 3 ? 1 : 3;
        const unsigned v = block_sad(use_win, dx, dy, off);
        if (v < sad) { sad = v; bx = off; *osel = (unsigned)q; }
      }
      mx = (int16_t)(m.x + ((s * bx) << 2));
    } else
      sad = block_sad(use_win, dx, dy, 0);
    const unsigned cost = (sad >> sh) + mv_cost(lam, m.y - mvp.y, mx - mvp.x);
    return valid ? cost : ~0u;
  };
  // min over the lanes [lo, lo + n) of (cost << 8 | lane - lo): the first candidate in evaluation order among the cheapest; ~0u for n == 0
  auto range_min = [&](unsigned cost, int lo, int n) -> unsigned {
    const int c = t.rank - lo;
    unsigned k = (cost << 8) | (unsigned)(c & 0xff);
    if (c < 0 || c >= n || cost == ~0u) k = ~0u;
    return team_min32(t, k);
  };
  struct NoWide { enum { value = 0 }; };
  struct Wide { enum { value = 1 }; };
  auto grid_mv = [&](mv_t centre, int step, int c) -> mv_t {   // point c of the 5x5 grid of spacing `step` around centre; the centre is skipped after the first step
    const int idx = (step < 32 && c >= 12) ? c + 1 : c;
    const int q = mul24(idx, 13) >> 6;   // idx / 5
    return mk_mv(centre.x + mul24(idx - q - (q << 2) - 2, step), centre.y + mul24(q - 2, step));
  };
  auto take = [&](mv_t m) {   // new optimum (clipped the way its candidate was), wave-uniform
    m = clip_mv(m, ypos, xpos, fw, fh, cb, cb, sign);
    mv_opt = mk_mv(tk_uniform(m.x), tk_uniform(m.y));
  };
  // --- telescope (encode_block.c:529-561): steps of 32, 16, 8, 4 quarter-pels.  A step is evaluated TOGETHER with the next one around the same
  // centre (25 + 24 or 24 + 24 lanes): when the step leaves the optimum on its centre - the usual case with a good predictor - the next step's grid is
  // exactly that one and its costs are already there; otherwise they are dropped and the next step runs from its real centre.
  for (int step = 32; step >= 4;) {
#ifdef TK_ME_NOSPEC   // tools/ubench_me.cpp: every step / round a pass of its own
    const int n1 = step < 32 ? 24 : 25, n2 = 0;
#else
    const int n1 = step < 32 ? 24 : 25, n2 = step > 4 ? 24 : 0;
#endif
    const mv_t centre = mv_ref;
    const int c1 = t.rank, c2 = t.rank - n1;
    const int v1 = c1 < n1, v2 = c2 >= 0 && c2 < n2;
    const mv_t m = v2 ? grid_mv(centre, step >> 1, c2) : grid_mv(centre, step, v1 ? c1 : 0);
    const unsigned cost = lane_cost(m, v1 || v2, NoWide(), nullptr);
    const unsigned k1 = range_min(cost, 0, n1);
    if (k1 != ~0u && (k1 >> 8) < min_sad) { min_sad = k1 >> 8; take(grid_mv(centre, step, (int)(k1 & 0xffu))); }
    mv_ref = mv_opt;
    step >>= 1;
    if (n2 && mv_ref.x == centre.x && mv_ref.y == centre.y) {   // the next step's centre is this one's: its costs are in lanes n1 .. n1 + 23
      const unsigned k2 = range_min(cost, n1, n2);
      if (k2 != ~0u && (k2 >> 8) < min_sad) { min_sad = k2 >> 8; take(grid_mv(centre, step, (int)(k2 & 0xffu))); }
      mv_ref = mv_opt;
      step >>= 1;
    }
  }
  // --- candidate list (encode_block.c:564-581)
  {
    const int n = TKU(lists->mvcand_num[ref_idx]);
    if (n > 0) {
      auto list_mv = [&](int c) -> mv_t { return mk_mv((int16_t)(lists->mvcand[ref_idx][c].x << 2), (int16_t)(lists->mvcand[ref_idx][c].y << 2)); };
      const int valid = t.rank < n;
      const mv_t m = list_mv(valid ? t.rank : 0);
      if (cb == 16) {
        unsigned osel = 0;
        const unsigned cost = lane_cost(m, valid, Wide(), &osel);
        const unsigned k = range_min(cost, 0, n);
        if (k != ~0u && (k >> 8) < min_sad) {
          min_sad = k >> 8;
          const int c = (int)(k & 0xffu);
          const int q = team_read_lane(t, (int)osel, c);   // the winner's offset
          mv_t mm = clip_mv(list_mv(c), ypos, xpos, fw, fh, cb, cb, sign);
          const int bx = q == 0 ? -3 : q == 1 ? -1 : q == 2 ? 0 : q == 3 ? 1 : 3;
          mm.x = (int16_t)(mm.x + ((s * bx) << 2));
          mv_opt = mk_mv(tk_uniform(mm.x), tk_uniform(mm.y));   // (the moved vector is not clipped again: encode_block.c:447-451)
        }
      } else {
        const unsigned cost = lane_cost(m, valid, NoWide(), nullptr);
        const unsigned k = range_min(cost, 0, n);
        if (k != ~0u && (k >> 8) < min_sad) { min_sad = k >> 8; take(list_mv((int)(k & 0xffu))); }
      }
    }
    mv_ref = mv_opt;
  }
  // --- hexagon refinement (encode_block.c:583-616): up to 5 rounds of 6, then 3 points.  A round is evaluated together with the next round of
  // every direction it can move in (6 + 6 x 3 or 3 + 3 x 3 lanes): the usual search ends after one or two rounds = one pass.
  {
    auto hex_off = [&](int dir, int* ox, int* oy) {
      *ox = dir == 0 ? 1 : dir == 1 ? 2 : dir == 2 ? 1 : dir == 3 ? -1 : dir == 4 ? -2 : -1;
      *oy = dir == 0 ? -1 : dir == 1 ? 0 : dir == 2 ? 1 : dir == 3 ? 1 : dir == 4 ? 0 : -1;
    };
    int start = 0, end = 5;
    for (int round = 1; round < 6;) {
      const int n = (end - start + 6) % 6 + 1;   // 6 in the first round, 3 afterwards
      const mv_t centre = mv_ref;
      // lanes [0, n): this round; lanes [n + 3 j, n + 3 j + 3): the next round if this one moves to its point j (new start = that direction - 1)
      const int L = t.rank;
      int ox, oy, valid = L < n * 4;
      mv_t m;
      {
        const int j = L < n ? L : mul24(L - n, 11) >> 5;   // (L - n) / 3 for L - n < 32
        const int dir1 = (start + (j < n ? j : 0)) % 6;
        hex_off(dir1, &ox, &oy);
        m = mk_mv(centre.x + ox * 4, centre.y + oy * 4);
        if (L >= n) {
          const int st2 = dir1 ? dir1 - 1 : 5;
          const int dir2 = (st2 + (L - n - mul24(j, 3))) % 6;
          hex_off(dir2, &ox, &oy);
          m = mk_mv(m.x + ox * 4, m.y + oy * 4);
        }
      }
#ifdef TK_ME_NOSPEC
      const int speculate = 0;
#else
      const int speculate = round < 5;
#endif
      if (!speculate) valid = L < n;
      const unsigned cost = lane_cost(m, valid, NoWide(), nullptr);
      int which = -1;
      const unsigned k = range_min(cost, 0, n);
      auto hex_mv = [&](mv_t ctr, int st, int c) -> mv_t { int x, y; hex_off((st + c) % 6, &x, &y); return mk_mv(ctr.x + x * 4, ctr.y + y * 4); };
      if (k != ~0u && (k >> 8) < min_sad) { min_sad = k >> 8; which = (int)(k & 0xffu); take(hex_mv(centre, start, which)); }
      int best_dir = which < 0 ? -1 : (start + which) % 6;
      // (the next round's centre is the point as this round evaluated it; a clipped point is not the speculated centre: fall back to a fresh pass)
      const mv_t raw = which < 0 ? centre : hex_mv(centre, start, which);
      mv_ref = mv_opt;
      const int start0 = start;
      start = best_dir ? best_dir - 1 : 5;
      end = start + 2;
      end -= (end >= 6) * 6;
      round++;
      if (best_dir < 0) break;
      if (speculate && round < 6 && raw.x == mv_ref.x && raw.y == mv_ref.y) {
        // the next round around the new centre: its three points are lanes n + 3 * which ..
        (void)start0;
        const mv_t centre2 = mv_ref;
        const unsigned k2 = range_min(cost, n + 3 * which, 3);
        int which2 = -1;
        if (k2 != ~0u && (k2 >> 8) < min_sad) { min_sad = k2 >> 8; which2 = (int)(k2 & 0xffu); take(hex_mv(centre2, start, which2)); }
        best_dir = which2 < 0 ? -1 : (start + which2) % 6;
        mv_ref = mv_opt;
        start = best_dir ? best_dir - 1 : 5;
        end = start + 2;
        end -= (end >= 6) * 6;
        round++;
        if (best_dir < 0) break;
      }
    }
  }
  return ((unsigned long long)min_sad << 32) | ((unsigned long long)(uint16_t)mv_opt.x << 16) | (unsigned long long)(uint16_t)mv_opt.y;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// One sub-pel pass (the eight half- or quarter-pel neighbours of `base`, encode_block.c:628-663) of an 8-bit PU of up to 32x32 samples with
// EIGHT LANES PER CANDIDATE (round 5).  tools/ubench_me.cpp: the generic pass costs ~8.5 k cycles for a 4x4 or 8x8 PU - nine luma_setups,
// eight tap tables and eight vector prices formed by every lane (~900 wave-instructions) around ~200 instructions of sample work.  Here lane
// (c, p) = candidate c = lane / 8, part p = lane % 8: a lane sets up, interpolates and prices ITS candidate only.  The PU is cut into
// column strips of 8 (4 for 4-row PUs) samples; a strip needs the 13 (9) window rows around it once: per row two v_dot4 on the eight
// bytes as loaded give the horizontal sum, six 24-bit multiply-adds per sample the vertical one (the strip form of subk8_strip_dy).
// The (1/2, 1/2) position's 12-tap filter (inter_prediction.c:146-160) is the same machinery with two horizontal tap sets (rows 1, 4:
// {0,0,1,1,0,0}; rows 2, 3: {0,1,2,2,1,0}), vertical weights {0,1,1,1,1,0} and rounding (sum + 8) >> 4, so the lanes of different
// candidates do not diverge; the second tap set is only formed when some candidate of the pass is such a position (wave-uniform).
// Requires every candidate's interpolation window inside the staged LDS window; returns 0xffffffff (the caller runs the generic pass) otherwise,
// else min over the candidates of (cost << 8 | c), cost exactly motion_estimate's.
template <int SP>
TK_DEVNI unsigned me_cand8_subpel(const Team t, const uint8_t* org_, int a_ostride, int a_width, int a_height, int a_sign, int a_fw, int a_fh, int a_xpos, int a_ypos,
                                  int a_bipred, double a_lam, const uint32_t* win_w32, int win_ox, int win_oy, int win_Ww, int win_Wh, int win_pitch, int win_on, mv_t base,
                                  int d, mv_t mvp) {
  enum : unsigned { kNone = 0xffffffffu };   // "not in the window": the caller runs the generic pass (a real key is below it: costs fit 24 bits)
  struct { int ostride, width, height, sign, fwidth, fheight, xpos, ypos, enable_bipred; double lam; } a_in = {a_ostride, a_width, a_height, a_sign, a_fw, a_fh, a_xpos, a_ypos, a_bipred, a_lam};
  MeWin win_in;
  win_in.w32 = win_w32; win_in.ox = win_ox; win_in.oy = win_oy; win_in.Ww = win_Ww; win_in.Wh = win_Wh; win_in.pitch = win_pitch; win_in.on = win_on;
  const int ostride = tk_uniform(a_in.ostride), width = tk_uniform(a_in.width), height = tk_uniform(a_in.height), sign = tk_uniform(a_in.sign);
  const int fw = tk_uniform(a_in.fwidth), fh = tk_uniform(a_in.fheight), xpos = tk_uniform(a_in.xpos), ypos = tk_uniform(a_in.ypos);
  const int bip = tk_uniform(a_in.enable_bipred);
  const double lam = tk_uniform_f64(a_in.lam);
  MeWin win;
  win.w32 = tk_uniform_ptr(win_in.w32); win.ox = tk_uniform(win_in.ox); win.oy = tk_uniform(win_in.oy); win.Ww = tk_uniform(win_in.Ww);
  win.Wh = tk_uniform(win_in.Wh); win.pitch = tk_uniform(win_in.pitch); win.on = tk_uniform(win_in.on);
  base = mk_mv(tk_uniform(base.x), tk_uniform(base.y));
  mvp = mk_mv(tk_uniform(mvp.x), tk_uniform(mvp.y));
  d = tk_uniform(d);
  org_ = tk_uniform_ptr(org_);
  if (!win.on) return kNone;
  const int c = t.rank >> 3, part = t.rank & 7;
  // order: (0,-d) (-d,0) (d,0) (0,d) (-d,-d) (-d,d) (d,-d) (d,d) as (y,x)
  const int oy = c == 0 ? 0 : c == 1 ? -d : c == 2 ? d : c == 3 ? 0 : c == 4 ? -d : c == 5 ? -d : d;
  const int ox = c == 0 ? -d : c == 1 ? 0 : c == 2 ? 0 : c == 3 ? d : c == 4 ? -d : c == 5 ? d : c == 6 ? -d : d;
  const mv_t mv = mk_mv(base.x + ox, base.y + oy);
  const SubPel sp = luma_setup(mv, sign, width, height, fw, fh, xpos, ypos, bip);
  const int centre = sp.ver_frac == 2 && sp.hor_frac == 2 && bip < 2;
  // interpolation window of the whole PU for this candidate: rows ver_int - 2 .. ver_int + height + 2, columns hor_int - 2 .. hor_int + width + 5
  const int outside = !(sp.hor_int - 2 >= win.ox && sp.hor_int + width + 6 <= win.ox + win.Ww && sp.ver_int - 2 >= win.oy && sp.ver_int + height + 3 <= win.oy + win.Wh);
  if (team_ballot(t, outside) != 0ull) return kNone;
  const int dual = team_ballot(t, centre) != 0ull;   // wave-uniform
  // per-lane filter description (see the header): horizontal taps A (vertical positions 0, 1, 4, 5) and B (2, 3) as int8 lanes, vertical weights
  const unsigned long long thA = centre ? 0x0000000001010000ull : sp.ph, thB = centre ? 0x0000000102020100ull : sp.ph;
  const int biasA = centre ? 128 * 2 : 128 * 64, biasB = centre ? 128 * 6 : 128 * 64;
  int tv[6];
  for (int m = 0; m < 6; m++) tv[m] = centre ? (m >= 1 && m <= 4 ? 1 : 0) : sp.tv[m];
  const int rnd = centre ? 8 : 2048, rsh = centre ? 4 : 12;
  const int lgw = ilog2((unsigned)width);
  const int SH = height == 4 ? 4 : 8;                      // strip height
  const int units = width * (height == 4 ? 1 : (height >> 3));
  unsigned sad = 0;
  auto strip = [&](auto sh_tag, auto dual_tag, int i0, int j) {
    constexpr int SHC = decltype(sh_tag)::value, DUAL = decltype(dual_tag)::value, NR = SHC + 5;
    const int woff = mul24(i0 + sp.ver_int - 2 - win.oy, win.pitch) + (j + sp.hor_int - 2 - win.ox);
    int hA[NR], hB[NR];
#if !TK_HOST
#pragma unroll
#endif
    for (int r = 0; r < NR; r++) {
      const Seg16 sg = win_seg<8>(win.w32, woff + mul24(r, win.pitch));
      const unsigned lo = sg.d[0] ^ 0x80808080u, hi = sg.d[1] ^ 0x80808080u;   // samples - 128 as int8 lanes
      hA[r] = dot4_i8((int)(unsigned)thA, (int)lo, dot4_i8((int)(unsigned)(thA >> 32), (int)hi, biasA));
      if constexpr (DUAL) hB[r] = dot4_i8((int)(unsigned)thB, (int)lo, dot4_i8((int)(unsigned)(thB >> 32), (int)hi, biasB));
      else hB[r] = hA[r];
    }
#if !TK_HOST
#pragma unroll
#endif
    for (int q = 0; q < SHC; q++) {
      int sum = mul24(tv[0], hA[q]) + mul24(tv[1], hA[q + 1]) + mul24(tv[2], hB[q + 2]) + mul24(tv[3], hB[q + 3]) + mul24(tv[4], hA[q + 4]) + mul24(tv[5], hA[q + 5]);
      const int pr = sat_pix((sum + rnd) >> rsh, 8);
      const int o = (int)spc<SP>(org_)[mul24(i0 + q, ostride) + j];
      sad += (unsigned)(o > pr ? o - pr : pr - o);
    }
  };
  struct S4 { enum { value = 4 }; };
  struct S8 { enum { value = 8 }; };
  struct D0 { enum { value = 0 }; };
  struct D1 { enum { value = 1 }; };
  for (int u = part; u < units; u += 8) {
    const int j = u & (width - 1), i0 = (u >> lgw) << 3;
    if (SH == 4) { if (dual) strip(S4(), D1(), 0, j); else strip(S4(), D0(), 0, j); }
    else { if (dual) strip(S8(), D1(), i0, j); else strip(S8(), D0(), i0, j); }
  }
  const unsigned tot = (unsigned)team_group_sum(t, (int)sad, 8);
  const unsigned cost = tot + mv_cost(lam, mv.y - mvp.y, mv.x - mvp.x);
  unsigned k = (cost << 8) | (unsigned)c;
  if (part != 0) k = ~0u;
  return team_min32(t, k);
}


// The same pass on 16-bit samples (round 6): lane (c, p) = candidate c = lane / 8, part p = lane % 8; a strip is one column of 8 (4) samples whose 13 (9)
// window rows are read once - six samples = three dwords at the candidate's byte offset (win_seg: any alignment), the six horizontal taps packed in
// pairs: three v_dot2_i32_i16 per row sum (|sum| <= 94 * 4095 < 2^19), six 24-bit multiply-adds per sample vertically; the (1/2, 1/2) position's 12-tap
// filter as two horizontal tap sets with vertical weights {0,1,1,1,1,0} and (sum + 8) >> 4, exactly as in me_cand8_subpel.  SAD >> (bitdepth - 8).
template <int SP>
TK_DEVNI unsigned me_cand16_subpel(const Team t, const uint16_t* org_, int a_ostride, int a_width, int a_height, int a_sign, int a_fw, int a_fh, int a_xpos, int a_ypos,
                                   int a_bipred, double a_lam, const uint32_t* win_w32, int win_ox, int win_oy, int win_Ww, int win_Wh, int win_pitch, int win_on, mv_t base,
                                   int d, mv_t mvp, int a_bitdepth) {
  enum : unsigned { kNone = 0xffffffffu };
  struct { int ostride, width, height, sign, fwidth, fheight, xpos, ypos, enable_bipred, bitdepth; double lam; } a_in = {a_ostride, a_width, a_height, a_sign, a_fw, a_fh, a_xpos, a_ypos, a_bipred, a_bitdepth, a_lam};
  MeWin win_in;
  win_in.w32 = win_w32; win_in.ox = win_ox; win_in.oy = win_oy; win_in.Ww = win_Ww; win_in.Wh = win_Wh; win_in.pitch = win_pitch; win_in.on = win_on;
  const int ostride = tk_uniform(a_in.ostride), width = tk_uniform(a_in.width), height = tk_uniform(a_in.height), sign = tk_uniform(a_in.sign);
  const int fw = tk_uniform(a_in.fwidth), fh = tk_uniform(a_in.fheight), xpos = tk_uniform(a_in.xpos), ypos = tk_uniform(a_in.ypos);
  const int bip = tk_uniform(a_in.enable_bipred), bitdepth = tk_uniform(a_in.bitdepth);
  const double lam = tk_uniform_f64(a_in.lam);
  MeWin win;
  win.w32 = tk_uniform_ptr(win_in.w32); win.ox = tk_uniform(win_in.ox); win.oy = tk_uniform(win_in.oy); win.Ww = tk_uniform(win_in.Ww);
  win.Wh = tk_uniform(win_in.Wh); win.pitch = tk_uniform(win_in.pitch); win.on = tk_uniform(win_in.on);
  base = mk_mv(tk_uniform(base.x), tk_uniform(base.y));
  mvp = mk_mv(tk_uniform(mvp.x), tk_uniform(mvp.y));
  d = tk_uniform(d);
  org_ = tk_uniform_ptr(org_);
  if (!win.on) return kNone;
  const int c = t.rank >> 3, part = t.rank & 7;
  const int oy = c == 0 ? 0 : c == 1 ? -d : c == 2 ? d : c == 3 ? 0 : c == 4 ? -d : c == 5 ? -d : d;
  const int ox = c == 0 ? -d : c == 1 ? 0 : c == 2 ? 0 : c == 3 ? d : c == 4 ? -d : c == 5 ? d : c == 6 ? -d : d;
  const mv_t mv = mk_mv(base.x + ox, base.y + oy);
  const SubPel sp = luma_setup(mv, sign, width, height, fw, fh, xpos, ypos, bip);
  const int centre = sp.ver_frac == 2 && sp.hor_frac == 2 && bip < 2;
  const int outside = !(sp.hor_int - 2 >= win.ox && sp.hor_int + width + 6 <= win.ox + win.Ww && sp.ver_int - 2 >= win.oy && sp.ver_int + height + 3 <= win.oy + win.Wh);
  if (team_ballot(t, outside) != 0ull) return kNone;
  const int dual = team_ballot(t, centre) != 0ull;   // wave-uniform
  auto pair = [](int a, int b) -> uint32_t { return (uint32_t)(uint16_t)(int16_t)a | ((uint32_t)(uint16_t)(int16_t)b << 16); };
  uint32_t tA[3], tB[3];
  for (int q = 0; q < 3; q++) {
    tA[q] = centre ? (q == 1 ? pair(1, 1) : 0u) : pair(sp.th[2 * q], sp.th[2 * q + 1]);                                        // rows 0, 1, 4, 5: {0,0,1,1,0,0}
    tB[q] = centre ? (q == 0 ? pair(0, 1) : q == 1 ? pair(2, 2) : pair(1, 0)) : pair(sp.th[2 * q], sp.th[2 * q + 1]);         // rows 2, 3:       {0,1,2,2,1,0}
  }
  int tv[6];
  for (int m = 0; m < 6; m++) tv[m] = centre ? (m >= 1 && m <= 4 ? 1 : 0) : sp.tv[m];
  const int rnd = centre ? 8 : 2048, rsh = centre ? 4 : 12;
  const int lgw = ilog2((unsigned)width);
  const int SH = height == 4 ? 4 : 8;
  const int units = width * (height == 4 ? 1 : (height >> 3));
  unsigned sad = 0;
  auto strip = [&](auto sh_tag, auto dual_tag, int i0, int j) {
    constexpr int SHC = decltype(sh_tag)::value, DUAL = decltype(dual_tag)::value, NR = SHC + 5;
    const int woff = mul24(i0 + sp.ver_int - 2 - win.oy, win.pitch) + ((j + sp.hor_int - 2 - win.ox) << 1);   // bytes
    int hA[NR], hB[NR];
#if !TK_HOST
#pragma unroll
#endif
    for (int r = 0; r < NR; r++) {
      const Seg16 sg = win_seg<12>(win.w32, woff + mul24(r, win.pitch));   // six samples
      hA[r] = dot2_i16(tA[0], sg.d[0], dot2_i16(tA[1], sg.d[1], dot2_i16(tA[2], sg.d[2], 0)));
      if constexpr (DUAL) hB[r] = dot2_i16(tB[0], sg.d[0], dot2_i16(tB[1], sg.d[1], dot2_i16(tB[2], sg.d[2], 0)));
      else hB[r] = hA[r];
    }
#if !TK_HOST
#pragma unroll
#endif
    for (int q = 0; q < SHC; q++) {
      int sum = mul24(tv[0], hA[q]) + mul24(tv[1], hA[q + 1]) + mul24(tv[2], hB[q + 2]) + mul24(tv[3], hB[q + 3]) + mul24(tv[4], hA[q + 4]) + mul24(tv[5], hA[q + 5]);
      const int pr = sat_pix((sum + rnd) >> rsh, bitdepth);
      const int o = (int)spc<SP>(org_)[mul24(i0 + q, ostride) + j];
      sad += (unsigned)(o > pr ? o - pr : pr - o);
    }
  };
  struct S4 { enum { value = 4 }; };
  struct S8 { enum { value = 8 }; };
  struct D0 { enum { value = 0 }; };
  struct D1 { enum { value = 1 }; };
  for (int u = part; u < units; u += 8) {
    const int j = u & (width - 1), i0 = (u >> lgw) << 3;
    if (SH == 4) { if (dual) strip(S4(), D1(), 0, j); else strip(S4(), D0(), 0, j); }
    else { if (dual) strip(S8(), D1(), i0, j); else strip(S8(), D0(), i0, j); }
  }
  const unsigned tot = (unsigned)team_group_sum(t, (int)sad, 8);
  const unsigned cost = (tot >> (bitdepth - 8)) + mv_cost(lam, mv.y - mvp.y, mv.x - mvp.x);
  unsigned k = (cost << 8) | (unsigned)c;
  if (part != 0) k = ~0u;
  return team_min32(t, k);
}


// Bilinear sub-pel approximations of encoder_speed > 0 (sad_calc_fasthalf enc/encode_block.c:174-283 ==
// sad_calc_fasthalf_simd enc_kernels.c:330, sad_calc_fastquarter :286-415): the SADs of the 8 half-
// (quarter-) pel neighbours of the centre built from rounding (avg) and truncating (rdavg) byte averages;
// returns the smallest of them and its offset.  Lanes split the samples, 8 shuffle reductions.
template <int SP, typename PIX> TK_DEV unsigned fast_halfpel(const Team t, const PIX* a_, const PIX* b, int as, int bs, int width, int height, int* bx, int* by) {
  const auto a = spc<SP>(a_);
  int tl = 0, tr = 0, br = 0, bl = 0, top = 0, right = 0, down = 0, left = 0;
  const Pow2 dw = mk_pow2(width);
  for (int r = t.rank; r < width * height; r += t.size) {
    int i, j;
    split2(dw, r, i, j);
    const PIX* c = b + i * bs + j;
    auto B = [&](int dy, int dx) -> int { return (int)c[dy * bs + dx]; };
    auto av = [](int x, int y) { return (x + y + 1) >> 1; };
    auto rd = [](int x, int y) { return (x + y) >> 1; };
    const int o = (int)a[i * as + j];
    const int h_l = av(B(0, -1), B(0, 0)), h_r = av(B(0, 0), B(0, 1));
    const int v4 = av(B(-2, 0), B(1, 0));          // column j,   rows -2 / +1
    const int v4b = av(B(-1, 0), B(2, 0));         // column j,   rows -1 / +2
    const int t6 = av(B(0, -2), B(0, 1));          // row 0, cols -2 / +1
    const int t7 = av(B(0, -1), B(0, 2));          // row 0, cols -1 / +2
    const int ptl = rd(rd(rd(av(B(-2, -1), B(1, -1)), v4), rd(av(B(-1, -2), B(-1, 1)), t6)), rd(av(B(-1, -1), B(-1, 0)), h_l));
    const int ptr = rd(rd(rd(v4, av(B(-2, 1), B(1, 1))), rd(t7, av(B(-1, -1), B(-1, 2)))), rd(av(B(-1, 0), B(-1, 1)), h_r));
    const int pbl = rd(rd(rd(v4b, av(B(-1, -1), B(2, -1))), rd(t6, av(B(1, -2), B(1, 1)))), rd(av(B(1, -1), B(1, 0)), h_l));
    const int pbr = rd(rd(rd(v4b, av(B(-1, 1), B(2, 1))), rd(t7, av(B(1, -1), B(1, 2)))), rd(h_r, av(B(1, 0), B(1, 1))));
    left += iabs(o - h_l); right += iabs(o - h_r);
    down += iabs(o - av(B(0, 0), B(1, 0))); top += iabs(o - av(B(0, 0), B(-1, 0)));
    tl += iabs(o - ptl); tr += iabs(o - ptr); br += iabs(o - pbr); bl += iabs(o - pbl);
  }
  unsigned utop = (unsigned)team_sum(t, top), uright = (unsigned)team_sum(t, right), udown = (unsigned)team_sum(t, down), uleft = (unsigned)team_sum(t, left);
  unsigned utl = (unsigned)team_sum(t, tl), utr = (unsigned)team_sum(t, tr), ubr = (unsigned)team_sum(t, br), ubl = (unsigned)team_sum(t, bl);
  int x = 0, y = -2;
  if (udown < utop) { y = 2; utop = udown; }
  if (uright < utop) { x = 2; y = 0; utop = uright; }
  if (uleft < utop) { x = -2; y = 0; utop = uleft; }
  if (utl < utop) { x = -2; y = -2; utop = utl; }
  if (utr < utop) { x = 2; y = -2; utop = utr; }
  if (ubr < utop) { x = 2; y = 2; utop = ubr; }
  if (ubl < utop) { x = -2; y = 2; utop = ubl; }
  *bx = x; *by = y;
  return utop;
}

template <int SP, typename PIX> TK_DEV unsigned fast_quarterpel(const Team t, const PIX* o__, const PIX* r_, int os, int rs, int width, int height, int* bx, int* by) {
  const auto o_ = spc<SP>(o__);
  int tl = 0, tr = 0, br = 0, bl = 0, top = 0, right = 0, down = 0, left = 0;
  const int hx = *bx, hy = *by;  // half-pel offset chosen before (0 or +-2): selects the interpolation pattern
  const Pow2 dw = mk_pow2(width);
  for (int q = t.rank; q < width * height; q += t.size) {
    int i, j;
    split2(dw, q, i, j);
    const PIX* c = r_ + i * rs + j;
    auto av = [](int x, int y) { return (x + y + 1) >> 1; };
    const int o = (int)o_[i * os + j];
    const int a = c[0], d = c[1], f = c[rs];
    int p_tl, p_top, p_tr, p_left, p_right, p_bl, p_down, p_br;
    if (hx & hy) {
      const int e = c[rs + 1];
      const int ad = av(a, d), de = av(d, e), af = av(a, f), fe = av(f, e);
      p_tl = (ad + af) >> 1; p_top = (de + a) >> 1; p_tr = (ad + de) >> 1; p_left = (ad + f) >> 1; p_right = (ad + e) >> 1;
      p_bl = (af + fe) >> 1; p_down = (de + f) >> 1; p_br = (de + fe) >> 1;
    } else if (hx) {
      const int b = c[-rs], cc = c[-rs + 1], e = c[rs + 1];
      const int ad = av(a, d), de = av(d, e), dc = av(d, cc), af = av(a, f), ab = av(a, b);
      p_tl = (ad + ab) >> 1; p_top = (dc + a) >> 1; p_tr = (ad + dc) >> 1; p_left = (ad + a) >> 1; p_right = (ad + d) >> 1;
      p_bl = (ad + af) >> 1; p_down = (af + d) >> 1; p_br = (ad + de) >> 1;
    } else if (hy) {
      const int e = c[rs + 1], g = c[rs - 1], h = c[-1];
      const int ad = av(a, d), af = av(a, f), fe = av(f, e), ah = av(a, h), gf = av(g, f);
      p_tl = (ah + af) >> 1; p_top = (af + a) >> 1; p_tr = (ad + af) >> 1; p_left = (gf + a) >> 1; p_right = (ad + f) >> 1;
      p_bl = (af + gf) >> 1; p_down = (af + f) >> 1; p_br = (af + fe) >> 1;
    } else {
      const int b = c[-rs], h = c[-1];
      const int ad = av(a, d), af = av(a, f), ah = av(a, h), ab = av(a, b);
      p_tl = (ah + ab) >> 1; p_top = (ab + a) >> 1; p_tr = (ad + ab) >> 1; p_left = (ah + a) >> 1; p_right = (ad + a) >> 1;
      p_bl = (ah + af) >> 1; p_down = (af + a) >> 1; p_br = (af + ad) >> 1;
    }
    tl += iabs(o - p_tl); top += iabs(o - p_top); tr += iabs(o - p_tr); left += iabs(o - p_left); right += iabs(o - p_right);
    bl += iabs(o - p_bl); down += iabs(o - p_down); br += iabs(o - p_br);
  }
  unsigned utop = (unsigned)team_sum(t, top), uright = (unsigned)team_sum(t, right), udown = (unsigned)team_sum(t, down), uleft = (unsigned)team_sum(t, left);
  unsigned utl = (unsigned)team_sum(t, tl), utr = (unsigned)team_sum(t, tr), ubr = (unsigned)team_sum(t, br), ubl = (unsigned)team_sum(t, bl);
  int x = 0, y = -1;
  if (utl < utop) { x = -1; utop = utl; }
  if (utr < utop) { x = 1; utop = utr; }
  if (uleft < utop) { x = -1; y = 0; utop = uleft; }
  if (uright < utop) { x = 1; y = 0; utop = uright; }
  if (ubl < utop) { x = -1; y = 1; utop = ubl; }
  if (udown < utop) { x = 0; y = 1; utop = udown; }
  if (ubr < utop) { x = 1; y = 1; utop = ubr; }
  *bx = x; *by = y;
  return utop;
}

// Stage ONE window for the coding block at (cb_x, cb_y) of size cb in reference `ref_idx` (ref_cb = its co-located sample in the padded
// plane), centred on the rounded search centre mvc: the HOR / VER / QUAD searches of this reference (eight motion_estimate calls) all start
// from it and find their window in LDS (motion_estimate: use_cb_win).  No window (cwin_valid = 0) when it does not fit the wave's budget
// with a reach of at least kMeWinRmin or would leave the padded plane.  The caller clears cwin_valid before the transform workspace is
// used again.
template <typename PIX>
TK_DEV void me_stage_cb_window(const Team t, MeWs* w_, const PIX* ref_cb, int rstride, int cb_x, int cb_y, int cb, mv_t mvc, int sign, int fwidth,
                               int fheight, int ref_idx) {
  const auto w = ldsc(w_);
  const int S = (int)sizeof(PIX), s = sign ? -1 : 1;
  const int cap = TKU(w->win_cap);
  int R = kMeWinR;
  while (R >= kMeWinRmin && me_win_bytes(cb, cb, R, S) > cap) R -= 4;
  const mv_t mv_ref = mk_mv(((mvc.x + 2) >> 2) << 2, ((mvc.y + 2) >> 2) << 2);
  const int Ww = cb + 2 * R, Wh = cb + 2 * R, pitch = Ww * S + 4;
  const int ox = s * (mv_ref.x >> 2) - R, oy = s * (mv_ref.y >> 2) - R;
  const int ok = TKU(w->win != nullptr && R >= kMeWinRmin && cb_x + ox >= -kPadY && cb_x + ox + Ww <= fwidth + kPadY && cb_y + oy >= -kPadY &&
                     cb_y + oy + Wh <= fheight + kPadY);
  t.sync();
  if (ok) {
    const uint32_t* w32 = lds_ld(&w_->win);
    const int rowb = Ww * S, spr = (rowb + 15) >> 4, total = spr * Wh;
    for (int k0 = 0; k0 < total; k0 += t.size) {
      const int k = k0 + t.rank;
      if (k < total) {
        const int row = k / spr, sg = k - row * spr;
        const Seg16 v = seg_load<SP_GLOBAL, 16>((const char*)(ref_cb + (oy + row) * rstride + ox) + 16 * sg);
        const int d = (row * pitch + 16 * sg) >> 2;
        const int nd = tmin(4, (rowb - 16 * sg) >> 2);
#if TK_HOST
        for (int q = 0; q < nd; q++) ((uint32_t*)w32)[d + q] = v.d[q];
#else
        TK_LDS uint32_t* l = (TK_LDS uint32_t*)(uint32_t)(uintptr_t)w32 + d;
        l[0] = v.d[0];
        if (nd > 1) l[1] = v.d[1];
        if (nd > 2) l[2] = v.d[2];
        if (nd > 3) l[3] = v.d[3];
#endif
      }
    }
  }
  if (t.rank == 0) {
    w->cwin_valid = ok; w->cwin_ref = ref_idx; w->cwin_ax = cb_x + ox; w->cwin_ay = cb_y + oy; w->cwin_Ww = Ww; w->cwin_Wh = Wh; w->cwin_pitch = pitch;
  }
  t.sync();
}

// SP: address space of the original-sample block `org` (LDS copy for coding blocks up to kLdsBlk and their 2*org-pred
// blocks, frame plane / global scratch above); w always lives in LDS on the device.
template <typename PIX, int SP>
TK_DEVNI unsigned motion_estimate(const Team t, MeWs* w_, const PIX* org, const PIX* ref, const MeArgs& a_in, mv_t mvc,
                                mv_t mvp, int ref_idx, mv_t* mv_out) {
  TK_PROF_T0();
#if defined(THOR_PROF_ME) && defined(THOR_PROF) && !TK_HOST
  const long long pme0_ = (long long)__builtin_readcyclecounter();
#endif
  const auto w = ldsc(w_);
  const auto lists = ldsc(lds_ld(&w_->lists));
  const auto orgs = spc<SP>(org);
  auto cmv_get = [&](int c) -> mv_t { return mk_mv(w->cmv[c].x, w->cmv[c].y); };
  auto cmv_set = [&](int c, mv_t m) { w->cmv[c].x = m.x; w->cmv[c].y = m.y; };
  MeArgs a_u;
  a_u.cb_size = tk_uniform(a_in.cb_size); a_u.ostride = tk_uniform(a_in.ostride); a_u.width = tk_uniform(a_in.width);
  a_u.height = tk_uniform(a_in.height); a_u.rstride = tk_uniform(a_in.rstride); a_u.sign = tk_uniform(a_in.sign);
  a_u.fwidth = tk_uniform(a_in.fwidth); a_u.fheight = tk_uniform(a_in.fheight); a_u.xpos = tk_uniform(a_in.xpos);
  a_u.ypos = tk_uniform(a_in.ypos); a_u.enable_bipred = tk_uniform(a_in.enable_bipred); a_u.bitdepth = tk_uniform(a_in.bitdepth);
  a_u.speed = tk_uniform(a_in.speed); a_u.lam = tk_uniform_f64(a_in.lam);
  a_u.pu_x = tk_uniform(a_in.pu_x); a_u.pu_y = tk_uniform(a_in.pu_y);
  const MeArgs& a = a_u;
  mvc = mk_mv(tk_uniform(mvc.x), tk_uniform(mvc.y));
  mvp = mk_mv(tk_uniform(mvp.x), tk_uniform(mvp.y));
  ref_idx = tk_uniform(ref_idx);
  const int s = a.sign ? -1 : 1;
  const int sh = a.bitdepth - 8;
  unsigned min_sad = kCostInit;
  mv_t mv_opt = mk_mv(0, 0);
  mv_t mv_ref = mk_mv(((mvc.x + 2) >> 2) << 2, ((mvc.y + 2) >> 2) << 2);
  // (rate: the vector's rate term, formed with the candidate - before its samples are waited for - not after the SAD)
  struct FP { mv_t mv; const PIX* p; int dx, dy; unsigned rate; };
  // (A table in LDS for the rate term - it is a function of a small bit count and of a per-frame constant - was measured SLOWER by 13 % per
  // call, tools/ubench_me.cpp / profiles/r05_ubench_me.md: the range check is a wave vote + branch per candidate, which serialises the four
  // candidate sets of an evaluator iteration; the double-precision chain pipelines across them.)
  auto rate_of = [&](mv_t m) -> unsigned { return mv_cost(a.lam, m.y - mvp.y, m.x - mvp.x); };
  auto fp_cost = [&](const FP& x, int sad) -> unsigned { return ((unsigned)sad >> sh) + x.rate; };
  // clip_mv leaves every vector within +-R quarter-pels of `ctr` alone when the block displaced by any of them stays inside
  // the padded area (one wave-uniform test per pass instead of four clamps per candidate; conservative for the
  // truncating division of clip_mv)
  auto clip_free = [&](mv_t ctr, int R) -> int {
    const int ext = kPadY - 16, cy = s * ctr.y, cx = s * ctr.x;
    return a.ypos + ((cy - R) >> 2) >= -ext && a.ypos + ((cy + R + 3) >> 2) + a.cb_size <= a.fheight + ext &&
           a.xpos + ((cx - R) >> 2) >= -ext && a.xpos + ((cx + R + 3) >> 2) + a.cb_size <= a.fwidth + ext;
  };
  auto mk_fp = [&](mv_t mv, int noclip) -> FP {
    FP x;
    x.mv = noclip ? mv : clip_mv(mv, a.ypos, a.xpos, a.fwidth, a.fheight, a.cb_size, a.cb_size, a.sign);
    x.dx = s * (x.mv.x >> 2);
    x.dy = s * (x.mv.y >> 2);
    x.p = ref + mul24(x.dy, a.rstride) + x.dx;
    x.rate = rate_of(x.mv);
    return x;
  };
#if defined(THOR_PROF) && !TK_HOST && !defined(THOR_PROF_NOMACROS)
  long long pq_ = (long long)__builtin_readcyclecounter();
  if (t.rank == 0) w->prof[11] += 1;
#endif
  // --- stage the search window in LDS (see MeWin): Wh rows of Ww bytes around the search centre with 16-byte row loads.  Only
  // when the whole window lies inside the padded reference plane (otherwise every pass of this search reads the plane).
  MeWin win;
  win.on = 0; win.w32 = nullptr; win.ox = win.oy = win.Ww = win.Wh = win.pitch = 0;
  const int use_cb_win = TKU(w->cwin_valid && w->cwin_ref == ref_idx && a.speed == 0);
  if (use_cb_win) {   // the block's window is already in LDS (me_stage_cb_window): this PU's view of it
    win.Ww = TKU(w->cwin_Ww); win.Wh = TKU(w->cwin_Wh); win.pitch = TKU(w->cwin_pitch);
    win.ox = TKU(w->cwin_ax) - a.pu_x; win.oy = TKU(w->cwin_ay) - a.pu_y;
    win.w32 = lds_ld(&w_->win);
    win.on = 1;
  } else {
    const int S = (int)sizeof(PIX);
    const int cap = TKU(w->win_cap);
    int R = kMeWinR;
    while (R >= kMeWinRmin && me_win_bytes(a.width, a.height, R, S) > cap) R -= 4;   // wave-uniform
    if (w->win && R >= kMeWinRmin && a.speed == 0) {
      win.Ww = a.width + 2 * R; win.Wh = a.height + 2 * R; win.pitch = win.Ww * S + 4;
      win.ox = s * (mv_ref.x >> 2) - R; win.oy = s * (mv_ref.y >> 2) - R;
      win.w32 = lds_ld(&w_->win);
      win.on = TKU(a.pu_x + win.ox >= -kPadY && a.pu_x + win.ox + win.Ww <= a.fwidth + kPadY && a.pu_y + win.oy >= -kPadY &&
                   a.pu_y + win.oy + win.Wh <= a.fheight + kPadY);
    }
    if (win.on) {
      const int rowb = win.Ww * S;   // a multiple of 4
      const int spr = (rowb + 15) >> 4, total = spr * win.Wh;   // 16-byte segments per row; the last one may read past the row (inside the plane's allocation)
      t.sync();
      for (int k0 = 0; k0 < total; k0 += t.size) {
        const int k = k0 + t.rank;
        if (k < total) {
          const int row = k / spr, sg = k - row * spr;
          const Seg16 v = seg_load<SP_GLOBAL, 16>((const char*)(ref + (win.oy + row) * a.rstride + win.ox) + 16 * sg);
          const int d = (row * win.pitch + 16 * sg) >> 2;
          const int nd = tmin(4, (rowb - 16 * sg) >> 2);   // dwords of this segment that belong to the row
#if TK_HOST
          for (int q = 0; q < nd; q++) ((uint32_t*)win.w32)[d + q] = v.d[q];
#else
          TK_LDS uint32_t* l = (TK_LDS uint32_t*)(uint32_t)(uintptr_t)win.w32 + d;
          l[0] = v.d[0];
          if (nd > 1) l[1] = v.d[1];
          if (nd > 2) l[2] = v.d[2];
          if (nd > 3) l[3] = v.d[3];
#endif
        }
      }
      t.sync();
    }
  }
  // 5-offset "widesad" evaluation of the clipped candidates w->cmv[0..n) (encode_block.c:430-453): per
  // candidate the offset with the smallest SAD (ties -> leftmost), then the usual cost with the adjusted
  // mv (written back to w->cmv).  Returns the best (cost << 32 | index).
  auto eval_wide = [&](int n) -> unsigned long long {
    unsigned long long bestk = ~0ull;
    for (int base = 0; base < n; base += kMeWideChunk) {
      const int m = n - base < kMeWideChunk ? n - base : kMeWideChunk;
      auto widepel = [&](int c5) -> FP {
        int c = mul24(c5, 13) >> 6, o = c5 - c - (c << 2);   // c5 / 5 for c5 < 60
        int off = o == 0 ? -3 : o == 1 ? -1 : o == 2 ? 0 : o == 3 ? 1 : 3;
        FP x;
        x.mv = cmv_get(base + c);
        x.dx = s * (x.mv.x >> 2) + off;
        x.dy = s * (x.mv.y >> 2);
        x.p = ref + mul24(x.dy, a.rstride) + x.dx;
        x.rate = 0;
        return x;
      };
      {
        const auto sadl = ldsc(w_->sad);
        seg_sads<SP>(t, m * 5, org, a.ostride, a.rstride, a.width, a.height, win, widepel, [&](int c5, const FP&, int sad, int mine) { if (mine) sadl[c5] = sad; });
      }
      t.sync();
      unsigned long long k = ~0ull;
      for (int lc = t.rank; lc < m; lc += t.size) {
        const int c = base + lc;
        mv_t mm = cmv_get(c);
        int x = 0;
        unsigned best = 1u << 31;
        for (int o = 0; o < 5; o++) {
          unsigned v = (unsigned)w->sad[lc * 5 + o];
          if (v < best) { best = v; x = o == 0 ? -3 : o == 1 ? -1 : o == 2 ? 0 : o == 3 ? 1 : 3; }
        }
        mm.x = (int16_t)(mm.x + ((s * x) << 2));
        cmv_set(c, mm);  // adjusted mv, looked up again if this candidate wins
        unsigned long long kk = ((unsigned long long)((best >> sh) + mv_cost(a.lam, mm.y - mvp.y, mm.x - mvp.x)) << 32) | (unsigned)c;   // (lanes diverge here: no table)
        k = kk < k ? kk : k;
      }
      t.sync();
      k = TKU64(team_min64(t, k));
      bestk = k < bestk ? k : bestk;
    }
    return bestk;
  };
  // Vectors this search has already evaluated cannot win a later pass: every pass keeps a candidate only on a strict '<'
  // against min_sad, the minimum over everything evaluated so far with the same cost function (same block, same mvp; a
  // vector is clipped the same way whenever it comes up).  The 5x5 grid of a telescope step (spacing `step`) shares its
  // points with even offsets with the previous step's grid (spacing 2*step) - up to 8 of 24; by induction the previous
  // grid is the only one that needs checking.  Passes of PUs that need more than one evaluator iteration for 24 candidates
  // leave those points out (encoder_speed 0; exact: the surviving candidates keep their relative order, so ties resolve the
  // same way); a hexagon refinement that starts on the centre of the last grid would only revisit it and is skipped.
  mv_t g_ctr = mk_mv(0, 0);
  int g_step = 0;   // spacing of the last telescope grid evaluated (0: none)
  // all four grids, for candidates that are not grid points themselves (the per-SB candidate list): a list entry that lies on
  // any of them has been evaluated
  mv_t gc32 = mk_mv(0, 0), gc16 = mk_mv(0, 0), gc8 = mk_mv(0, 0);   // centres of the grids of step 32, 16, 8 (step 4: g_ctr); named
                                                                      // scalars, not an array: nothing here is indexed at run time
  auto on_grid_of = [&](mv_t m, mv_t ctr, int gs) -> int {
    const int dx = m.x - ctr.x, dy = m.y - ctr.y;
    return !((dx | dy) & (gs - 1)) && iabs(dx) <= 2 * gs && iabs(dy) <= 2 * gs;
  };
  auto on_any_grid = [&](mv_t m) -> int {
    // g_step == 4: the telescope has run (all four steps, encoder_speed 0) - the four centres are this search's
    return g_step == 4 && (on_grid_of(m, gc32, 32) | on_grid_of(m, gc16, 16) | on_grid_of(m, gc8, 8) | on_grid_of(m, g_ctr, 4));
  };
  auto on_grid = [&](mv_t m) -> int {
    const int dx = m.x - g_ctr.x, dy = m.y - g_ctr.y;
    return g_step && !((dx | dy) & (g_step - 1)) && iabs(dx) <= 2 * g_step && iabs(dy) <= 2 * g_step;
  };
  const int lw_ = (a.width < 16 / (int)sizeof(PIX)) ? a.width : 16 / (int)sizeof(PIX);   // samples per row segment (seg_sads)
  const int dedup = a.speed == 0 && a.height * (a.width / lw_) >= 16;
  // PUs of up to 32x32 samples, encoder_speed 0: the one-lane-per-candidate full-pel search (me_cand_fullpel; 16-bit samples since round 6) - same passes, same result
  int small_done = 0;
  {
    // (rows of 64 and 128 samples keep the 64-lane evaluator: a lane walking 256+ row segments of the plane by itself measured 2x slower, profiles/r05_ubench_me.md)
    if (TKU(a.speed == 0 && t.size == 64 && a.width <= 32 && a.height <= 32)) {
#ifndef TK_ME_NO_SMALL
      unsigned long long fr;
#define TK_ME_FP_ARGS t, w_, org, ref, a.cb_size, a.ostride, a.width, a.height, a.rstride, a.sign, a.fwidth, a.fheight, a.xpos, a.ypos, a.lam, win.w32, win.ox, win.oy, win.Ww, win.Wh, win.pitch, win.on, mvc, mvp, ref_idx, sh
      const int rowb = a.width * (int)sizeof(PIX);   // bytes per row: the segment size
      if (rowb == 4) { if constexpr (sizeof(PIX) == 1) fr = me_cand_fullpel<PIX, 4, SP>(TK_ME_FP_ARGS); else fr = 0; }
      else if (rowb == 8) fr = me_cand_fullpel<PIX, 8, SP>(TK_ME_FP_ARGS);
      else fr = me_cand_fullpel<PIX, 16, SP>(TK_ME_FP_ARGS);
#undef TK_ME_FP_ARGS
      min_sad = (unsigned)(fr >> 32);
      mv_opt = mk_mv((int16_t)(uint16_t)(fr >> 16), (int16_t)(uint16_t)fr);
      mv_ref = mv_opt;
      small_done = 1;
#endif
    }
  }
#ifdef TK_ME_CROSSCHECK   // test builds: run the generic passes as well and stop the kernel when the two searches disagree
  const int xs_have = small_done;
  const unsigned xs_min = min_sad;
  const mv_t xs_mv = mv_opt;
  if (small_done) { small_done = 0; min_sad = kCostInit; mv_opt = mk_mv(0, 0); mv_ref = mk_mv(((mvc.x + 2) >> 2) << 2, ((mvc.y + 2) >> 2) << 2); }
#endif
  if (!small_done) {
  // --- telescope (encode_block.c:529-561); encoder_speed > 0 keeps it only for 16x16 CBs with bipred on
  if ((a.cb_size == 16 && a.enable_bipred) || a.speed == 0)
  for (int step = 32; step >= 4; step >>= 1) {
    const int n = step < 32 ? 24 : 25;
    const mv_t centre = mv_ref;
    const int noclip = TKU(clip_free(centre, 2 * step));
    auto tele_mv = [&](int c) -> mv_t {
      int idx = (step < 32 && c >= 12) ? c + 1 : c;  // centre skipped after the first step
      int q = mul24(idx, 13) >> 6;                     // idx / 5 for idx < 25 (24-bit multiplies: full rate, v_mul_lo_u32 is a quarter)
      return mk_mv(centre.x + mul24(idx - q - (q << 2) - 2, step), centre.y + mul24(q - 2, step));
    };
    auto tele = [&](int c) -> FP { return mk_fp(tele_mv(c), noclip); };
    if (dedup && g_step) {
      // compact list of the grid points not evaluated before (clipped) in w->cmv, in grid order
      t.sync();
      int cnt = 0;
      for (int c0 = 0; c0 < n; c0 += t.size) {
        const int c = c0 + t.rank;
        const mv_t m = tele_mv(c < n ? c : 0);
        const int keep = c < n && !on_grid(m);
        const unsigned long long mask = team_ballot(t, keep);
        if (keep) cmv_set(cnt + __builtin_popcountll(mask & ((1ull << t.rank) - 1ull)), mk_fp(m, noclip).mv);
        cnt += __builtin_popcountll(mask);
      }
      t.sync();
      cnt = TKU(cnt);
      auto cl = [&](int c) -> FP { return mk_fp(cmv_get(c), 1); };
      unsigned long long k = eval_fullpel<SP>(t, cnt, org, a.ostride, a.rstride, a.width, a.height, win, cl, fp_cost);
      if ((unsigned)(k >> 32) < min_sad) { min_sad = (unsigned)(k >> 32); mv_opt = cmv_get((int)(unsigned)k); }
      t.sync();
    } else
    if (step == 32 && a.cb_size == 16 && a.speed == 1) {  // first ring by widesad at encoder_speed 1
      t.sync();
      for (int c = t.rank; c < n; c += t.size) cmv_set(c, tele(c).mv);
      t.sync();
      unsigned long long k = eval_wide(n);
      if ((unsigned)(k >> 32) < min_sad) { min_sad = (unsigned)(k >> 32); mv_opt = cmv_get((int)(unsigned)k); }
      t.sync();
    } else {
      unsigned long long k = eval_fullpel<SP>(t, n, org, a.ostride, a.rstride, a.width, a.height, win, tele, fp_cost);
      if ((unsigned)(k >> 32) < min_sad) { min_sad = (unsigned)(k >> 32); mv_opt = tele((int)(unsigned)k).mv; }
    }
    g_ctr = centre; g_step = a.speed == 0 ? step : 0;
    if (step == 32) gc32 = centre; else if (step == 16) gc16 = centre; else if (step == 8) gc8 = centre;
    mv_ref = mv_opt;
  }

#if defined(THOR_PROF) && !TK_HOST && !defined(THOR_PROF_NOMACROS)
  if (t.rank == 0) w->prof[13] += (long long)__builtin_readcyclecounter() - pq_;
  pq_ = (long long)__builtin_readcyclecounter();
#endif
  // --- candidate list (encode_block.c:564-581)
  {
    const int n = TKU(lists->mvcand_num[ref_idx]);
    if (n > 0) {
      const int wide = a.cb_size == 16;
      for (int c = t.rank; c < n; c += t.size) {
        const mv_t m = mk_mv(lists->mvcand[ref_idx][c].x, lists->mvcand[ref_idx][c].y);
        cmv_set(c, clip_mv(mk_mv(m.x << 2, m.y << 2), a.ypos, a.xpos, a.fwidth, a.fheight, a.cb_size, a.cb_size, a.sign));
      }
      t.sync();
      if (wide) {
        unsigned long long bestk = eval_wide(n);
        if ((unsigned)(bestk >> 32) < min_sad) { min_sad = (unsigned)(bestk >> 32); mv_opt = cmv_get((int)(unsigned)bestk); }
        t.sync();
      } else {
        // Entries that lie on one of the telescope grids have been evaluated (a clipped vector that coincides with a grid point
        // was evaluated as itself: clip_mv is idempotent): the list - previous results of this superblock, clustered around the
        // motion the telescope has just walked to - is compacted in place to the others, in list order (45-56 % go on the test
        // content); nothing left: no pass.
        int cnt = n;
        if (a.speed == 0) {
          cnt = 0;
          for (int c0 = 0; c0 < n; c0 += t.size) {
            const int c = c0 + t.rank;
            const mv_t m = cmv_get(c < n ? c : 0);
            const int keep = c < n && !on_any_grid(m);
            const unsigned long long mask = team_ballot(t, keep);
            t.sync();   // every lane holds its entry before lower slots are rewritten
            if (keep) cmv_set(cnt + __builtin_popcountll(mask & ((1ull << t.rank) - 1ull)), m);
            cnt += __builtin_popcountll(mask);
          }
          t.sync();
          cnt = TKU(cnt);
        }
        if (cnt > 0) {
          auto cl = [&](int c) -> FP { return mk_fp(cmv_get(c), 1); };  // cmv already clipped (clip_mv is idempotent)
          unsigned long long k = eval_fullpel<SP>(t, cnt, org, a.ostride, a.rstride, a.width, a.height, win, cl, fp_cost);
          if ((unsigned)(k >> 32) < min_sad) { min_sad = (unsigned)(k >> 32); mv_opt = cmv_get((int)(unsigned)k); }
        }
        t.sync();
      }
    }
    mv_ref = mv_opt;
  }

#if defined(THOR_PROF) && !TK_HOST && !defined(THOR_PROF_NOMACROS)
  if (t.rank == 0) w->prof[14] += (long long)__builtin_readcyclecounter() - pq_;
  pq_ = (long long)__builtin_readcyclecounter();
#endif
  // --- hexagon refinement (encode_block.c:583-616): up to 5 rounds; skipped for CBs > 16 at encoder_speed > 0
  {
    int start = 0, end = 5;
    // all six points around the centre of the last telescope grid (spacing one sample) belong to that grid: nothing to find
    const int revisit = TKU(g_step == 4 && mv_ref.x == g_ctr.x && mv_ref.y == g_ctr.y);
    const int maxsteps = revisit ? 0 : (a.cb_size <= 16 || a.speed == 0) ? 6 : 0;
    for (int step = 1; step < maxsteps; step++) {
      const int n = (end - start + 6) % 6 + 1;  // 6 in the first round, 3 afterwards
      const mv_t centre = mv_ref;
      const int noclip = TKU(clip_free(centre, 8));
      auto hex = [&](int c) -> FP {
        int dir = (start + c) % 6;
        int ox = dir == 0 ? 1 : dir == 1 ? 2 : dir == 2 ? 1 : dir == 3 ? -1 : dir == 4 ? -2 : -1;  // "diy" -> x
        int oy = dir == 0 ? -1 : dir == 1 ? 0 : dir == 2 ? 1 : dir == 3 ? 1 : dir == 4 ? 0 : -1;  // "dix" -> y
        return mk_fp(mk_mv(centre.x + ox * 4, centre.y + oy * 4), noclip);
      };
      int which = -1;
      unsigned long long k = eval_fullpel<SP>(t, n, org, a.ostride, a.rstride, a.width, a.height, win, hex, fp_cost);
      if ((unsigned)(k >> 32) < min_sad) { min_sad = (unsigned)(k >> 32); which = (int)(unsigned)k; mv_opt = hex(which).mv; }
      int best_dir = which < 0 ? -1 : (start + which) % 6;
      mv_ref = mv_opt;
      start = best_dir ? best_dir - 1 : 5;
      end = start + 2;
      end -= (end >= 6) * 6;
      if (best_dir < 0) break;
    }
  }

#if defined(THOR_PROF) && !TK_HOST && !defined(THOR_PROF_NOMACROS)
  if (t.rank == 0) w->prof[15] += (long long)__builtin_readcyclecounter() - pq_;
#endif
  }   // !small_done
#ifdef TK_ME_CROSSCHECK
  if (xs_have && (xs_min != min_sad || xs_mv.x != mv_opt.x || xs_mv.y != mv_opt.y)) {
#if TK_HOST
    fprintf(stderr, "me_small8_fullpel disagrees with motion_estimate: %u (%d,%d) vs %u (%d,%d)\n", xs_min, xs_mv.x, xs_mv.y, min_sad, mv_opt.x, mv_opt.y);
    abort();
#else
    __builtin_trap();
#endif
  }
#endif
  TK_PROF_ADD(w, 2);
  // --- half-pel then quarter-pel (encode_block.c:628-663)
#if defined(THOR_PROF) && !TK_HOST && !defined(THOR_PROF_NOMACROS)
  pt0_ = (long long)__builtin_readcyclecounter();
#endif
  unsigned cmin = min_sad;
#ifdef TK_ME_NO_SUBPEL   // tools/ubench_me.cpp: the full-pel part alone
  if (false) {
#else
  {
#endif
  if (a.speed == 0)
  for (int pass = 0; pass < 2; pass++) {
    const int d = pass == 0 ? 2 : 1;
    const mv_t base = pass == 0 ? mv_ref : mv_opt;
#ifdef TK_ME_CROSSCHECK
    unsigned xs_sub = ~0u;   // result of me_cand8_subpel for this pass (test builds compare it with the generic pass)
#endif
    // order: (0,-d) (-d,0) (d,0) (0,d) (-d,-d) (-d,d) (d,-d) (d,d) as (y,x)
    struct SPc { mv_t mv; SubPel sp; };
    auto sub_prep = [&](int c) -> SPc {
      int oy = c == 0 ? 0 : c == 1 ? -d : c == 2 ? d : c == 3 ? 0 : c == 4 ? -d : c == 5 ? -d : d;
      int ox = c == 0 ? -d : c == 1 ? 0 : c == 2 ? 0 : c == 3 ? d : c == 4 ? -d : c == 5 ? d : c == 6 ? -d : d;
      SPc x;
      x.mv = mk_mv(base.x + ox, base.y + oy);
      x.sp = luma_setup(x.mv, a.sign, a.width, a.height, a.fwidth, a.fheight, a.xpos, a.ypos, a.enable_bipred);
      return x;
    };
    const Pow2 dw = mk_pow2(a.width);
    auto sub_item = [&](const SPc& x, int r) -> int {
      int i, j;
      split2(dw, r, i, j);
      return iabs((int)orgs[i * a.ostride + j] - luma_sample(ref, a.rstride, i, j, x.sp, a.enable_bipred, a.bitdepth));
    };
    auto sub_cost = [&](int, const SPc& x, int sad) -> unsigned {
      return ((unsigned)sad >> sh) + mv_cost(a.lam, x.mv.y - mvp.y, x.mv.x - mvp.x);
    };
    // PUs of up to 16x16 samples whose interpolation windows lie in the staged window: eight lanes per candidate (me_cand8_subpel; 16-bit samples since
    // round 6: me_cand16_subpel)
    {
#ifndef TK_ME_NO_SMALL
      if (TKU(a.width * a.height <= 256 && t.size == 64)) {   // (32x32: the 64-lane strip form below is faster - tools/ubench_me.cpp)
        unsigned k32;
        if constexpr (sizeof(PIX) == 1)
          k32 = me_cand8_subpel<SP>(t, org, a.ostride, a.width, a.height, a.sign, a.fwidth, a.fheight, a.xpos, a.ypos, a.enable_bipred, a.lam, win.w32, win.ox, win.oy,
                                    win.Ww, win.Wh, win.pitch, win.on, base, d, mvp);
        else
          k32 = me_cand16_subpel<SP>(t, org, a.ostride, a.width, a.height, a.sign, a.fwidth, a.fheight, a.xpos, a.ypos, a.enable_bipred, a.lam, win.w32, win.ox, win.oy,
                                     win.Ww, win.Wh, win.pitch, win.on, base, d, mvp, a.bitdepth);
        if (k32 != 0xffffffffu) {
#ifdef TK_ME_CROSSCHECK
          xs_sub = k32;
#else
          mv_t bestv = base;
          if ((k32 >> 8) < cmin) {
            cmin = k32 >> 8;
            const int c = (int)(k32 & 0xffu);
            const int oy = c == 0 ? 0 : c == 1 ? -d : c == 2 ? d : c == 3 ? 0 : c == 4 ? -d : c == 5 ? -d : d;
            const int ox = c == 0 ? -d : c == 1 ? 0 : c == 2 ? 0 : c == 3 ? d : c == 4 ? -d : c == 5 ? d : c == 6 ? -d : d;
            bestv = mk_mv(base.x + ox, base.y + oy);
          }
          mv_opt = mk_mv(mv_opt.x + (bestv.x - base.x), mv_opt.y + (bestv.y - base.y));
          continue;
#endif
        }
      }
#endif
    }
    // Fast path: all eight candidates read from the 8x8 window around the centre's integer position (always,
    // except when luma_setup's frame-edge clamps pull a candidate further away).
    TK_PROF_MARK(ps0_);
    SPc cand[8];
    const SubPel ctr = luma_setup(base, a.sign, a.width, a.height, a.fwidth, a.fheight, a.xpos, a.ypos, a.enable_bipred);
    int in_window = 1;
    for (int c = 0; c < 8; c++) {
      cand[c] = sub_prep(c);
      const int dy = cand[c].sp.ver_int - ctr.ver_int + 1, dx = cand[c].sp.hor_int - ctr.hor_int + 1;
      if (dy < 0 || dy > 2 || dx < 0 || dx > 2) in_window = 0;
    }
    unsigned long long k;
    (void)0;
    TK_PROF_MARK(ps1_);
#if defined(THOR_PROF_SUBPEL) && defined(THOR_PROF) && !TK_HOST   // tools/ubench_me.cpp: set-up / sample loop / reduction of a sub-pel pass
    if (t.rank == 0) w->prof[6] += ps1_ - ps0_;
#endif
    if (in_window) {
      int sad8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      // per-candidate parameters are identical in every lane: scalar registers
      SubK8 k8[8];
      for (int c = 0; c < 8; c++) {
        SubPel usp = cand[c].sp;   // (subk8_make reads the packed taps and the fractions only)
        usp.ph = tk_uniform64(usp.ph); usp.pv = tk_uniform64(usp.pv);
        usp.ver_frac = tk_uniform(usp.ver_frac); usp.hor_frac = tk_uniform(usp.hor_frac);
        k8[c] = subk8_make(usp, tk_uniform(cand[c].sp.ver_int - ctr.ver_int + 1), tk_uniform(cand[c].sp.hor_int - ctr.hor_int + 1), a.enable_bipred);
      }
      const int sub_in_win = TKU(win.on && ctr.hor_int - 3 >= win.ox && ctr.hor_int + a.width + 5 <= win.ox + win.Ww &&
                                 ctr.ver_int - 3 >= win.oy && ctr.ver_int + a.height + 5 <= win.oy + win.Wh);
      if (sizeof(PIX) == 1 && a.width * a.height >= 512) {
        // large PUs: a lane takes a vertical strip of eight samples of one column (tk_pred.h:subk8_strip)
        if constexpr (sizeof(PIX) == 1) {
          for (int r = t.rank; r < a.width * (a.height >> 3); r += t.size) {
            int st, j;
            split2(dw, r, st, j);
            const int i0 = st << 3;
            const PIX* p0 = ref + (i0 + ctr.ver_int - 3) * a.rstride + (j + ctr.hor_int - 3);
            unsigned long long wb[15];
            if (sub_in_win) {   // the rows come out of the staged window
              const int woff = (i0 + ctr.ver_int - 3 - win.oy) * win.pitch + (j + ctr.hor_int - 3 - win.ox);
#if !TK_HOST
#pragma unroll
#endif
              for (int q = 0; q < 15; q++) { const Seg16 sg = win_seg<8>(win.w32, woff + q * win.pitch); wb[q] = (((unsigned long long)sg.d[1] << 32) | sg.d[0]) ^ 0x8080808080808080ull; }
            } else {
#if !TK_HOST
#pragma unroll
#endif
              for (int q = 0; q < 15; q++) wb[q] = gload64(p0 + q * a.rstride) ^ 0x8080808080808080ull;
            }
            int o8[8];
#if !TK_HOST
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) o8[q] = (int)orgs[(i0 + q) * a.ostride + j];
#if !TK_HOST
#pragma unroll
#endif
            for (int c = 0; c < 8; c++) sad8[c] = subk8_strip(wb, k8[c], o8, sad8[c]);
          }
        }
      } else if (sizeof(PIX) == 2) {
        // 16-bit samples: eight window rows of eight samples per prediction sample (window or plane), candidates by v_dot2 (tk_pred.h:SubK16)
        if constexpr (sizeof(PIX) == 2) {
          SubK16 k16[8];
          for (int c = 0; c < 8; c++) {
            SubPel usp = cand[c].sp;
            for (int m = 0; m < 6; m++) { usp.tv[m] = tk_uniform(usp.tv[m]); usp.th[m] = tk_uniform(usp.th[m]); }
            usp.ver_frac = tk_uniform(usp.ver_frac); usp.hor_frac = tk_uniform(usp.hor_frac);
            k16[c] = subk16_make(usp, tk_uniform(cand[c].sp.ver_int - ctr.ver_int + 1), tk_uniform(cand[c].sp.hor_int - ctr.hor_int + 1), a.enable_bipred);
          }
          if (a.width * a.height >= 512) {
            // large PUs: a lane takes a vertical strip of eight samples of one column (tk_pred.h:subk16_strip)
            for (int r = t.rank; r < a.width * (a.height >> 3); r += t.size) {
              int st, j;
              split2(dw, r, st, j);
              const int i0 = st << 3;
              uint32_t wb[15][4];
              if (sub_in_win) {
                const int woff = (i0 + ctr.ver_int - 3 - win.oy) * win.pitch + (j + ctr.hor_int - 3 - win.ox) * 2;
#if !TK_HOST
#pragma unroll
#endif
                for (int q = 0; q < 15; q++) { const Seg16 sg = win_seg<16>(win.w32, woff + q * win.pitch); wb[q][0] = sg.d[0]; wb[q][1] = sg.d[1]; wb[q][2] = sg.d[2]; wb[q][3] = sg.d[3]; }
              } else {
                const PIX* p0 = ref + (i0 + ctr.ver_int - 3) * a.rstride + (j + ctr.hor_int - 3);
#if !TK_HOST
#pragma unroll
#endif
                for (int q = 0; q < 15; q++) { const Seg16 sg = seg_load<SP_GLOBAL, 16>(p0 + q * a.rstride); wb[q][0] = sg.d[0]; wb[q][1] = sg.d[1]; wb[q][2] = sg.d[2]; wb[q][3] = sg.d[3]; }
              }
              int o8[8];
#if !TK_HOST
#pragma unroll
#endif
              for (int q = 0; q < 8; q++) o8[q] = (int)orgs[(i0 + q) * a.ostride + j];
#if !TK_HOST
#pragma unroll
#endif
              for (int c = 0; c < 8; c++) sad8[c] = subk16_strip(wb, k16[c], o8, sad8[c], a.bitdepth);
            }
          } else
          for (int r = t.rank; r < a.width * a.height; r += t.size) {
            int i, j;
            split2(dw, r, i, j);
            uint32_t rows[8][4];
            if (sub_in_win) {
              const int woff = (i + ctr.ver_int - 3 - win.oy) * win.pitch + (j + ctr.hor_int - 3 - win.ox) * 2;
#if !TK_HOST
#pragma unroll
#endif
              for (int q = 0; q < 8; q++) { const Seg16 sg = win_seg<16>(win.w32, woff + q * win.pitch); rows[q][0] = sg.d[0]; rows[q][1] = sg.d[1]; rows[q][2] = sg.d[2]; rows[q][3] = sg.d[3]; }
            } else {
              const PIX* p0 = ref + (i + ctr.ver_int - 3) * a.rstride + (j + ctr.hor_int - 3);
#if !TK_HOST
#pragma unroll
#endif
              for (int q = 0; q < 8; q++) { const Seg16 sg = seg_load<SP_GLOBAL, 16>(p0 + q * a.rstride); rows[q][0] = sg.d[0]; rows[q][1] = sg.d[1]; rows[q][2] = sg.d[2]; rows[q][3] = sg.d[3]; }
            }
            const int o = (int)orgs[i * a.ostride + j];
#if !TK_HOST
#pragma unroll
#endif
            for (int c = 0; c < 8; c++) sad8[c] += iabs(o - subk16_sample(rows, k16[c], a.bitdepth));
          }
        }
      } else
      for (int r = t.rank; r < a.width * a.height; r += t.size) {
        int i, j;
        split2(dw, r, i, j);
        const PIX* p0 = ref + (i + ctr.ver_int - 3) * a.rstride + (j + ctr.hor_int - 3);
        WinRow<PIX> wr[8];
        if (sub_in_win) {   // the (PU + 8)^2 samples around the centre are inside the staged window
          if constexpr (sizeof(PIX) == 1) {
            const int woff = (i + ctr.ver_int - 3 - win.oy) * win.pitch + (j + ctr.hor_int - 3 - win.ox);
#if !TK_HOST
#pragma unroll
#endif
            for (int q = 0; q < 8; q++) { const Seg16 sg = win_seg<8>(win.w32, woff + q * win.pitch); wr[q].a = ((unsigned long long)sg.d[1] << 32) | sg.d[0]; }
          }
        } else
          for (int q = 0; q < 8; q++) win_load(p0 + q * a.rstride, wr[q]);
        const int o = (int)orgs[i * a.ostride + j];
        if constexpr (sizeof(PIX) == 1) {
          unsigned long long wb[8];
          for (int q = 0; q < 8; q++) wb[q] = wr[q].a ^ 0x8080808080808080ull;   // samples - 128 as int8 lanes
#if !TK_HOST
#pragma unroll
#endif
          for (int c = 0; c < 8; c++) {
            const unsigned pr = (unsigned)subk8_sample(wb, k8[c]), uo = (unsigned)o;
            sad8[c] += (int)((uo > pr ? uo : pr) - (uo < pr ? uo : pr));
          }
        } else {
          for (int c = 0; c < 8; c++) {
            const int dy = cand[c].sp.ver_int - ctr.ver_int + 1, dx = cand[c].sp.hor_int - ctr.hor_int + 1;
            WinRow<PIX> rows[6];
            for (int m = 0; m < 6; m++) rows[m] = win_pick(wr[m], wr[m + 1], wr[m + 2], dy, dx);
            sad8[c] += iabs(o - luma_sample_win<PIX>(rows, cand[c].sp, a.enable_bipred, a.bitdepth));
          }
        }
      }
      TK_PROF_ACC(w, 10, ps1_);
      TK_PROF_MARK(ps2_);
      k = ~0ull;
      for (int c = 0; c < 8; c++) {
        const int tot = team_sum(t, sad8[c]);
        const unsigned long long kk = ((unsigned long long)sub_cost(c, cand[c], tot) << 32) | (unsigned)c;
        k = kk < k ? kk : k;
      }
#if defined(THOR_PROF_SUBPEL) && defined(THOR_PROF) && !TK_HOST
      TK_PROF_ACC(w, 7, ps2_);
#endif
    } else
      k = eval_min(t, 8, a.width * a.height, sub_prep, sub_item, sub_cost);
#ifdef TK_ME_CROSSCHECK
    if (xs_sub != ~0u && ((xs_sub >> 8) != (unsigned)(k >> 32) || (xs_sub & 0xffu) != ((unsigned)k & 0xffu))) {
#if TK_HOST
      abort();
#else
      __builtin_trap();
#endif
    }
#endif
    mv_t best = base;
    if ((unsigned)(k >> 32) < cmin) { cmin = (unsigned)(k >> 32); best = sub_prep((int)(unsigned)k).mv; }
    // mv_opt += delta of the winning position (none => unchanged)
    mv_opt = mk_mv(mv_opt.x + (best.x - base.x), mv_opt.y + (best.y - base.y));
  }
  else {
    // bilinear approximation (encode_block.c:664-707).  NB the reference folds the sign into mv_ref before
    // pricing the half-pel vector, so for a backward reference the rate term sees the negated vector.
    mv_t mr = mk_mv(mv_ref.x * s, mv_ref.y * s);
    int spx = 0, spy = 0, xd_hp = 0, yd_hp = 0, xd_qp = 0, yd_qp = 0;
    unsigned sad = fast_halfpel<SP>(t, org, ref + (mr.y >> 2) * a.rstride + (mr.x >> 2), a.ostride, a.rstride, a.width, a.height, &spx, &spy) >> sh;
    sad += mv_cost(a.lam, mr.y + s * spy - mvp.y, mr.x + s * spx - mvp.x);
    if (sad < cmin) { cmin = sad; xd_hp = s * spx; yd_hp = s * spy; }
    spx = xd_hp; spy = yd_hp;
    mr = mk_mv(mv_opt.x + s * spx, mv_opt.y + s * spy);
    mv_opt = mk_mv(mv_opt.x + xd_hp, mv_opt.y + yd_hp);
    sad = fast_quarterpel<SP>(t, org, ref + (s * (mr.y >> 2)) * a.rstride + s * (mr.x >> 2), a.ostride, a.rstride, a.width, a.height, &spx, &spy) >> sh;
    sad += mv_cost(a.lam, mr.y + s * spy - mvp.y, mr.x + s * spx - mvp.x);
    if (sad < cmin) { cmin = sad; xd_qp = s * spx; yd_qp = s * spy; }
    mv_opt = mk_mv(mv_opt.x + xd_qp, mv_opt.y + yd_qp);
  }
  }
  TK_PROF_ADD(w, 3);
#if defined(THOR_PROF_ME) && defined(THOR_PROF) && !TK_HOST
  // whole-call cycles and call counts by coding-block size (this build's code_tu does not use slots 16..25)
  if (t.rank == 0) {
    const int cls = a.cb_size <= 8 ? 0 : a.cb_size == 16 ? 1 : a.cb_size == 32 ? 2 : a.cb_size == 64 ? 3 : 4;
    w->prof[16 + cls] += (long long)__builtin_readcyclecounter() - pme0_;
    w->prof[21 + cls] += 1;
  }
#endif
  *mv_out = mv_opt;
  return cmin < min_sad ? cmin : min_sad;
}


// motion_estimate_bi (enc/encode_block.c:798-914): joint search of ONE vector used as +mv on ref0 and
// -mv on ref1 (B frames, encoder_speed 0).  3x3 telescope from 8 px down to 1/4 px around the rounded
// centre, then six "extra" candidates read from the per-SB candidate list of r_idx0 - including the
// reference's side effect on that list (slots [num..3] zero-filled, slots 4 and 5 overwritten with mvp
// and (0,0) without touching the count) and its use of the list's full-pel entries
// as quarter-pel vectors.  The vector is clipped for ref0's sign and then AGAIN for ref1's sign; ref0 is
// predicted with the once-clipped vector, ref1 and the cost use the twice-clipped one.
// The search has two phases: the telescope (phase bit 1), which depends only on the block, the two reference planes, mvc and mvp,
// and the six extra candidates (phase bit 2), which read - and clobber - the candidate list as it stands after the bi-prediction
// search of the block.  The block decision of a B frame runs the telescope early on another wavefront (tk_block.h:MD_BIJOINT) and
// hands its result (min_sad_in, *mv_out) to the second phase; phase 3 = both, back to back.
template <typename PIX, int SP>
TK_DEVNI unsigned motion_estimate_bi(const Team t, MeWs* w_, const PIX* org_, const PIX* ref0, const PIX* ref1, const MeArgs& a,
                                    mv_t mvc, mv_t mvp, int r_idx0, mv_t* mv_out, int phase = 3, unsigned min_sad_in = kCostInit) {
  const auto org = spc<SP>(org_);
  const auto lists = ldsc(lds_ld(&w_->lists));
  const int sh = a.bitdepth - 8;
  const int size = a.cb_size;
  unsigned min_sad = (phase & 1) ? (unsigned)kCostInit : min_sad_in;
  mv_t mv_opt = (phase & 1) ? mk_mv(0, 0) : *mv_out;
  mv_t mv_ref = mk_mv(((mvc.x + 2) >> 2) << 2, ((mvc.y + 2) >> 2) << 2);
  struct BI { mv_t mv; SubPel s0, s1; };
  auto mk_bi = [&](mv_t mv) -> BI {
    BI x;
    mv_t m0 = clip_mv(mv, a.ypos, a.xpos, a.fwidth, a.fheight, size, size, a.sign);
    mv_t m1 = clip_mv(m0, a.ypos, a.xpos, a.fwidth, a.fheight, size, size, 1 - a.sign);
    x.mv = m1;
    x.s0 = luma_setup(m0, a.sign, size, size, a.fwidth, a.fheight, a.xpos, a.ypos, a.enable_bipred);
    x.s1 = luma_setup(m1, 1 - a.sign, size, size, a.fwidth, a.fheight, a.xpos, a.ypos, a.enable_bipred);
    return x;
  };
  const Pow2 dw = mk_pow2(size);
  auto bi_item = [&](const BI& x, int r) -> int {
    int i, j;
    split2(dw, r, i, j);
    int p0 = luma_sample(ref0, a.rstride, i, j, x.s0, a.enable_bipred, a.bitdepth);
    int p1 = luma_sample(ref1, a.rstride, i, j, x.s1, a.enable_bipred, a.bitdepth);
    return iabs((int)org[i * a.ostride + j] - ((p0 + p1) >> 1));
  };
  auto bi_cost = [&](int, const BI& x, int sad) -> unsigned {
    return ((unsigned)sad >> sh) + mv_cost(a.lam, (int16_t)(x.mv.y - mvp.y), (int16_t)(x.mv.x - mvp.x));
  };
  for (int step = (phase & 1) ? 32 : 0; step > 0; step >>= 1) {
    // candidate list of this step in the reference's (k outer = y, l inner = x) order
    int ox[9], oy[9], n = 0;
    const int vf = mv_ref.y & 3, hf = mv_ref.x & 3;
    for (int k = -step; k <= step; k += step)
      for (int l = -step; l <= step; l += step) {
        if (step < 32 && k == 0 && l == 0) continue;
        if (step == 1) {
          int skip;
          if (vf == 0 && hf == 0) skip = iabs(k) != iabs(l);
          else if (vf == 2 && hf == 2) skip = 1;
          else skip = iabs(k) == iabs(l);
          if (skip) continue;
        }
        ox[n] = l; oy[n] = k; n++;
      }
    if (n > 0) {
      const mv_t centre = mv_ref;
      auto cand = [&](int c) -> BI {
        int x = 0, y = 0;
        for (int q = 0; q < 9; q++) if (q == c) { x = ox[q]; y = oy[q]; }
        return mk_bi(mk_mv(centre.x + x, centre.y + y));
      };
      // A step whose candidates all sit on integer positions in both references (every step of 4 quarter-pels and more, unless a
      // frame-edge clamp of luma_setup interferes) predicts by copying: the SAD against the truncating average of the two displaced
      // blocks runs on the row-segment evaluator of the uni-directional search (16 bytes of each reference per lane and memory
      // instruction) instead of one sample per lane and step.
      // (the steps of 2 and 1 quarter-pels move off the integer grid: no need to build their candidates twice to find that out)
      int all_int = step >= 4;
      for (int c = 0; c < n && all_int; c++) { const BI x = cand(c); all_int = all_int && !(x.s0.ver_frac | x.s0.hor_frac | x.s1.ver_frac | x.s1.hor_frac); }
      unsigned long long k;
      if (TKU(all_int)) {
        struct FB { mv_t mv; const PIX* p; const PIX* p2; int dx, dy; };
        auto cand_fp = [&](int c) -> FB {
          const BI x = cand(c);
          FB f;
          f.mv = x.mv; f.dx = f.dy = 0;
          f.p = ref0 + x.s0.ver_int * a.rstride + x.s0.hor_int;
          f.p2 = ref1 + x.s1.ver_int * a.rstride + x.s1.hor_int;
          return f;
        };
        MeWin nowin;
        nowin.on = 0; nowin.w32 = nullptr; nowin.ox = nowin.oy = nowin.Ww = nowin.Wh = nowin.pitch = 0;
        k = eval_fullpel<SP>(t, n, org_, a.ostride, a.rstride, size, size, nowin, cand_fp,
                             [&](const FB& x, int sad) -> unsigned { return ((unsigned)sad >> sh) + mv_cost(a.lam, (int16_t)(x.mv.y - mvp.y), (int16_t)(x.mv.x - mvp.x)); });
      } else
        k = eval_min(t, n, size * size, cand, bi_item, bi_cost);
      if ((unsigned)(k >> 32) < min_sad) { min_sad = (unsigned)(k >> 32); mv_opt = cand((int)(unsigned)k).mv; }
    }
    mv_ref = mv_opt;
  }
  if (!(phase & 2)) { *mv_out = mv_opt; return min_sad; }
  // extra candidates (+ side effect on the shared list)
  t.sync();
  if (t.rank == 0) {
    for (int idx = lists->mvcand_num[r_idx0]; idx < 4; idx++) { lists->mvcand[r_idx0][idx].x = 0; lists->mvcand[r_idx0][idx].y = 0; }
    lists->mvcand[r_idx0][4].x = mvp.x; lists->mvcand[r_idx0][4].y = mvp.y;
    lists->mvcand[r_idx0][5].x = 0; lists->mvcand[r_idx0][5].y = 0;
  }
  t.sync();
  {
    auto cand = [&](int c) -> BI { return mk_bi(mk_mv(lists->mvcand[r_idx0][c].x, lists->mvcand[r_idx0][c].y)); };
    unsigned long long k = eval_min(t, 6, size * size, cand, bi_item, bi_cost);
    if ((unsigned)(k >> 32) < min_sad) { min_sad = (unsigned)(k >> 32); mv_opt = cand((int)(unsigned)k).mv; }
  }
  t.sync();
  *mv_out = mv_opt;
  return min_sad;
}

}  // namespace tk
