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

enum { kMeMaxCand = 64 * 5 };

struct MeWs {
  int sad[kMeMaxCand];
  unsigned cost[64];
  mv_t cmv[64];
  // per-SB candidate lists (frame_info.mvcand[], enc/mainenc.h:146-148), reset per SB
  mv_t mvcand[kMaxRefs][64];
  int mvcand_num[kMaxRefs];
  unsigned long long mvcand_mask[kMaxRefs];
  long long* prof;
};

TK_DEV int mv_len1(int a) {
  a = iabs(a);
  if (a < 1) return 2;
  if (a < 2) return 4;
  if (a < 4) return 5;
  if (a < 36) return 5 + ((a - 4) >> 3) + 1;
  return 10 + ((a - 36) >> 4) + 1;
}
TK_DEV int quote_mv_bits(int dy, int dx) { return mv_len1(dx) + mv_len1(dy); }
TK_DEV unsigned mv_cost(double lam, int dy, int dx) {
  return (unsigned)mul_add_nofma(lam, (double)quote_mv_bits(dy, dx), 0.5);
}

// add_mvcandidate (encode_block.c:69-82) - call from ONE lane.
TK_DEV void add_mvcand(MeWs* w, int r, mv_t mv) {
  mv_t imv = mk_mv((mv.x + 2) >> 2, (mv.y + 2) >> 2);
  unsigned long long m = 1ull << ((((int)imv.y << 3) ^ (int)imv.x) & 63);
  if (!(m & w->mvcand_mask[r])) {
    w->mvcand[r][w->mvcand_num[r]] = imv;
    w->mvcand_num[r] += 1;
  }
  w->mvcand_mask[r] |= m;
}

// sad[c] = sum over w x h of |org - F(c, i, j)| for c < ncand.
template <typename PIX, class F>
TK_DEV void sad_many(const Team& t, int* sad, int ncand, const PIX* org, int ostride, int w, int h, F sample) {
  for (int c = t.rank; c < ncand; c += t.size) sad[c] = 0;
  t.sync();
  const int npix = w * h;
  if (npix >= t.size) {
    for (int c = 0; c < ncand; c++) {
      int local = 0;
      for (int k = t.rank; k < npix; k += t.size) {
        int i = k / w, j = k - i * w;
        local += iabs((int)org[i * ostride + j] - sample(c, i, j));
      }
      team_add(&sad[c], local);
    }
  } else {
    for (int it = t.rank; it < ncand * npix; it += t.size) {
      int c = it / npix, k = it - c * npix;
      int i = k / w, j = k - i * w;
      team_add(&sad[c], iabs((int)org[i * ostride + j] - sample(c, i, j)));
    }
  }
  t.sync();
}

// |a-b| summed over 4 horizontally adjacent samples (a: 4-sample aligned, b: any alignment).
template <typename PIX> TK_DEV int sad4(const PIX* a, const PIX* b) {
  return iabs((int)a[0] - (int)b[0]) + iabs((int)a[1] - (int)b[1]) + iabs((int)a[2] - (int)b[2]) + iabs((int)a[3] - (int)b[3]);
}
#if !TK_HOST
template <> __device__ __forceinline__ int sad4<uint8_t>(const uint8_t* a, const uint8_t* b) {
  uint32_t va, vb;
  __builtin_memcpy(&va, a, 4);
  __builtin_memcpy(&vb, b, 4);
  return (int)__builtin_amdgcn_sad_u8(va, vb, 0u);  // v_sad_u8: 4 byte-SADs per lane-op
}
#endif

// sad[c] = SAD(org block, block at base(c)) for c < ncand; full-pel candidates given as pointers.
// Work item = (candidate, row, group of 4 samples).
template <typename PIX, class F>
TK_DEV void sad_many_ptr(const Team& t, int* sad, int ncand, const PIX* org, int ostride, int w, int h, int rstride, F base) {
  for (int c = t.rank; c < ncand; c += t.size) sad[c] = 0;
  t.sync();
  const int gpr = w >> 2, nit = h * gpr;
  if (nit >= t.size) {
    for (int c = 0; c < ncand; c++) {
      const PIX* b = base(c);
      int local = 0;
      for (int r = t.rank; r < nit; r += t.size) {
        int i = r / gpr, g = r - i * gpr;
        local += sad4(org + i * ostride + 4 * g, b + i * rstride + 4 * g);
      }
      team_add(&sad[c], local);
    }
  } else {
    for (int it = t.rank; it < ncand * nit; it += t.size) {
      int c = it / nit, r = it - c * nit;
      int i = r / gpr, g = r - i * gpr;
      team_add(&sad[c], sad4(org + i * ostride + 4 * g, base(c) + i * rstride + 4 * g));
    }
  }
  t.sync();
}

struct MeArgs {
  int cb_size;           // `size` argument of motion_estimate = CB size
  int ostride;           // stride of the original-sample block
  int width, height;     // PU dims
  int rstride;
  int sign, fwidth, fheight, xpos, ypos;  // CB position (Appendix B.16)
  int enable_bipred, bitdepth;
  double lam;            // sqrt(lambda)
};

// Pick the first strictly smaller cost in evaluation order (all lanes run this redundantly).
TK_DEV void pick_best(const MeWs* w, int n, unsigned& min_sad, mv_t& mv_opt, int* which = nullptr) {
  for (int c = 0; c < n; c++)
    if (w->cost[c] < min_sad) {
      min_sad = w->cost[c];
      mv_opt = w->cmv[c];
      if (which) *which = c;
    }
}

template <typename PIX>
TK_DEVNI unsigned motion_estimate(const Team& t, MeWs* w, const PIX* org, const PIX* ref, const MeArgs& a, mv_t mvc,
                                mv_t mvp, int ref_idx, mv_t* mv_out) {
  TK_PROF_T0();
  const int s = a.sign ? -1 : 1;
  const int sh = a.bitdepth - 8;
  unsigned min_sad = kCostInit;
  mv_t mv_opt = mk_mv(0, 0);
  mv_t mv_ref = mk_mv(((mvc.x + 2) >> 2) << 2, ((mvc.y + 2) >> 2) << 2);
  auto fullpel = [&](int c) -> const PIX* {
    mv_t m = w->cmv[c];
    return ref + (s * (m.y >> 2)) * a.rstride + s * (m.x >> 2);
  };

  // --- telescope (encode_block.c:529-561)
  for (int step = 32; step >= 4; step >>= 1) {
    const int range = 2 * step;
    int n = 0;
    // candidate index in evaluation order: k (y) outer, l (x) inner, centre skipped after step 32
    for (int idx = t.rank; idx < 25; idx += t.size) {
      int k = (idx / 5 - 2) * step, l = (idx % 5 - 2) * step;
      int slot = idx;
      if (step < 32) {
        if (idx == 12) continue;
        if (idx > 12) slot = idx - 1;
      }
      (void)range;
      w->cmv[slot] = clip_mv(mk_mv(mv_ref.x + l, mv_ref.y + k), a.ypos, a.xpos, a.fwidth, a.fheight, a.cb_size,
                             a.cb_size, a.sign);
    }
    n = step < 32 ? 24 : 25;
    t.sync();
    sad_many_ptr(t, w->sad, n, org, a.ostride, a.width, a.height, a.rstride, fullpel);
    for (int c = t.rank; c < n; c += t.size) {
      mv_t m = w->cmv[c];
      w->cost[c] = ((unsigned)w->sad[c] >> sh) + mv_cost(a.lam, m.y - mvp.y, m.x - mvp.x);
    }
    t.sync();
    pick_best(w, n, min_sad, mv_opt);
    t.sync();
    mv_ref = mv_opt;
  }

  // --- candidate list (encode_block.c:564-581)
  {
    const int n = w->mvcand_num[ref_idx];
    if (n > 0) {
      const int wide = a.cb_size == 16;
      for (int c = t.rank; c < n; c += t.size) {
        mv_t m = w->mvcand[ref_idx][c];
        w->cmv[c] = clip_mv(mk_mv(m.x << 2, m.y << 2), a.ypos, a.xpos, a.fwidth, a.fheight, a.cb_size, a.cb_size, a.sign);
      }
      t.sync();
      if (wide) {
        auto widepel = [&](int c5) -> const PIX* {
          int c = c5 / 5, o = c5 - c * 5;
          int off = o == 0 ? -3 : o == 1 ? -1 : o == 2 ? 0 : o == 3 ? 1 : 3;
          mv_t m = w->cmv[c];
          return ref + (s * (m.y >> 2)) * a.rstride + s * (m.x >> 2) + off;
        };
        sad_many_ptr(t, w->sad, n * 5, org, a.ostride, a.width, a.height, a.rstride, widepel);
      } else {
        sad_many_ptr(t, w->sad, n, org, a.ostride, a.width, a.height, a.rstride, fullpel);
      }
      for (int c = t.rank; c < n; c += t.size) {
        mv_t m = w->cmv[c];
        unsigned sad;
        int x = 0;
        if (wide) {
          unsigned best = 1u << 31;
          for (int o = 0; o < 5; o++) {
            unsigned v = (unsigned)w->sad[c * 5 + o];
            if (v < best) { best = v; x = o == 0 ? -3 : o == 1 ? -1 : o == 2 ? 0 : o == 3 ? 1 : 3; }
          }
          sad = best;
        } else sad = (unsigned)w->sad[c];
        sad >>= sh;
        m.x = (int16_t)(m.x + ((s * x) << 2));
        w->cost[c] = sad + mv_cost(a.lam, m.y - mvp.y, m.x - mvp.x);
        w->cmv[c] = m;
      }
      t.sync();
      pick_best(w, n, min_sad, mv_opt);
      t.sync();
    }
    mv_ref = mv_opt;
  }

  // --- hexagon refinement (encode_block.c:583-616), encoder_speed 0 => up to 5 rounds
  {
    int start = 0, end = 5;
    for (int step = 1; step < 6; step++) {
      const int n = (end - start + 6) % 6 + 1;  // 6 in the first round, 3 afterwards
      for (int c = t.rank; c < n; c += t.size) {
        int dir = (start + c) % 6;
        int ox = dir == 0 ? 1 : dir == 1 ? 2 : dir == 2 ? 1 : dir == 3 ? -1 : dir == 4 ? -2 : -1;  // "diy" -> x
        int oy = dir == 0 ? -1 : dir == 1 ? 0 : dir == 2 ? 1 : dir == 3 ? 1 : dir == 4 ? 0 : -1;  // "dix" -> y
        w->cmv[c] = clip_mv(mk_mv(mv_ref.x + ox * 4, mv_ref.y + oy * 4), a.ypos, a.xpos, a.fwidth, a.fheight,
                            a.cb_size, a.cb_size, a.sign);
      }
      t.sync();
      sad_many_ptr(t, w->sad, n, org, a.ostride, a.width, a.height, a.rstride, fullpel);
      for (int c = t.rank; c < n; c += t.size) {
        mv_t m = w->cmv[c];
        w->cost[c] = ((unsigned)w->sad[c] >> sh) + mv_cost(a.lam, m.y - mvp.y, m.x - mvp.x);
      }
      t.sync();
      int which = -1;
      pick_best(w, n, min_sad, mv_opt, &which);
      t.sync();
      int best_dir = which < 0 ? -1 : (start + which) % 6;
      mv_ref = mv_opt;
      start = best_dir ? best_dir - 1 : 5;
      end = start + 2;
      end -= (end >= 6) * 6;
      if (best_dir < 0) break;
    }
  }

  TK_PROF_ADD(w, 2);
  // --- half-pel then quarter-pel (encode_block.c:628-663)
#if defined(THOR_PROF) && !TK_HOST
  pt0_ = (long long)__builtin_readcyclecounter();
#endif
  unsigned cmin = min_sad;
  for (int pass = 0; pass < 2; pass++) {
    const int d = pass == 0 ? 2 : 1;
    const mv_t base = pass == 0 ? mv_ref : mv_opt;
    for (int c = t.rank; c < 8; c += t.size) {
      // order: (0,-d) (-d,0) (d,0) (0,d) (-d,-d) (-d,d) (d,-d) (d,d) as (y,x)
      int oy = c == 0 ? 0 : c == 1 ? -d : c == 2 ? d : c == 3 ? 0 : c == 4 ? -d : c == 5 ? -d : d;
      int ox = c == 0 ? -d : c == 1 ? 0 : c == 2 ? 0 : c == 3 ? d : c == 4 ? -d : c == 5 ? d : c == 6 ? -d : d;
      w->cmv[c] = mk_mv(base.x + ox, base.y + oy);
    }
    t.sync();
    auto subpel = [&](int c, int i, int j) -> int {
      SubPel sp = luma_setup(w->cmv[c], a.sign, a.width, a.height, a.fwidth, a.fheight, a.xpos, a.ypos);
      return luma_sample(ref, a.rstride, i, j, sp, a.enable_bipred, a.bitdepth);
    };
    sad_many(t, w->sad, 8, org, a.ostride, a.width, a.height, subpel);
    for (int c = t.rank; c < 8; c += t.size) {
      mv_t m = w->cmv[c];
      w->cost[c] = ((unsigned)w->sad[c] >> sh) + mv_cost(a.lam, m.y - mvp.y, m.x - mvp.x);
    }
    t.sync();
    mv_t best = base;
    pick_best(w, 8, cmin, best);
    t.sync();
    // mv_opt += delta of the winning position (none => unchanged)
    mv_opt = mk_mv(mv_opt.x + (best.x - base.x), mv_opt.y + (best.y - base.y));
  }
  TK_PROF_ADD(w, 3);
  *mv_out = mv_opt;
  return cmin < min_sad ? cmin : min_sad;
}

}  // namespace tk
