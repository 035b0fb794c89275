// tk_bits.h - bit sink (count or emit) and the block-level syntax of the Thor bitstream.
// Restates, for RDO bit counting on the device AND for the final emission, the syntax of
// enc/putvlc.c:73-160 (put_vlc), enc/write_bits.c:123-143 (write_mv), :145-241 (write_coeff),
// :257-358 (write_super_mode) and :360-600 (write_block).  4:2:0 only, no delta-QP syntax
// (max_delta_qp = bitrate = 0 in every BASELINE config).
#pragma once
#include "tk_common.h"

namespace tk {

struct BitSink {
  uint32_t* buf;  // word w holds stream bits [32w, 32w+32), first bit in the MSB
  int pos;        // bit position
  int cap;        // capacity in bits
  int emit;       // 0: count only
  int ovf;
  // emission state between bs_open() and bs_close(): the bits of the word being assembled (`fill` = pos & 31 of them, in the
  // low bits of `acc`).  Complete words are stored without reading the buffer back - the single-lane emission of a block
  // is a chain of register operations plus fire-and-forget stores instead of a global read-modify-write per syntax element.
  uint32_t acc = 0;
  int fill = 0;
  // cooperative emission (every lane of a team runs the same emission code on wave-uniform values): only the lane with store != 0
  // writes the words
  int store = 1;
};
// Start / finish emitting at b.pos (one lane).  Bits of the last word beyond the end position are unspecified (they always
// were: positions are rewound and re-emitted during the quadtree walk; consumers read pos bits).
TK_DEV void bs_open(BitSink& b) {
  b.fill = b.pos & 31;
  b.acc = (b.emit && b.fill && b.pos < b.cap) ? b.buf[b.pos >> 5] >> (32 - b.fill) : 0u;
}
TK_DEV void bs_close(BitSink& b) {
  if (b.emit && b.fill && !b.ovf && b.store) b.buf[b.pos >> 5] = b.acc << (32 - b.fill);
}

// E = false: counting only (the emission code is not even compiled into the caller: the counting instances are the ones
// every RDO trial runs).  E = true: between bs_open() and bs_close().
template <bool E> TK_DEV void bs_put_t(BitSink& b, int n, uint32_t val) {
  if (E && b.emit && n > 0) {
    if (b.pos + n > b.cap) {
      b.ovf = 1;
    } else {
      const uint32_t msk = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
      val &= msk;
      const int room = 32 - b.fill;
      if (n < room) {
        b.acc = (b.acc << n) | val;
        b.fill += n;
      } else {
        const int lo = n - room;  // bits that go to the next word
        if (b.store) b.buf[b.pos >> 5] = (room >= 32 ? 0u : (b.acc << room)) | (val >> lo);
        b.acc = lo ? (val & ((1u << lo) - 1u)) : 0u;
        b.fill = lo;
      }
    }
  }
  b.pos += n;
}
TK_DEV void bs_put(BitSink& b, int n, uint32_t val) { bs_put_t<true>(b, n, val); }

// (len, code) of VLC table n for symbol cn (enc/putvlc.c:73-160).
TK_DEV void vlc_code(int n, uint32_t cn, int& len, uint32_t& code) {
  if (n == 6 || n == 7) {
    if (cn == 0) { len = 2; code = 2; return; }
    if (n == 6) { cn++; n = 2; }
    else {
      if (cn == 1) { len = 3; code = 6; return; }
      if (cn < 4) { len = 4; code = (7u << 1) | (cn & 1u); return; }
      cn += 4; n = 3;
    }
  }
  if (n <= 5) {
    uint32_t t = 1u << n;
    if ((int)cn < (int)(5u * t)) {
      code = t + (cn & (t - 1u));
      len = 1 + n + (int)(cn >> n);
    } else {
      code = cn - 5u * t + t;
      len = (5 - n) + 1 + 2 * ilog2(code);
    }
    return;
  }
  if (n == 8) {
    if (cn < 6) { len = 2 + (int)(cn >> 1); code = 2u + (cn & 1u); }
    else { len = 5; code = cn - 6u; }
    return;
  }
  if (n == 10) { code = cn + 1u; len = 1 + 2 * ilog2(code); return; }
  // 11..18: truncated unary with m = n-10 symbols+1
  uint32_t m = (uint32_t)(n - 10);
  len = (cn == m) ? (int)m : (int)cn + 1;
  code = (cn != m) ? 1u : 0u;
}

template <bool E> TK_DEV int bs_vlc_t(BitSink& b, int n, uint32_t cn) {
  int len; uint32_t code;
  vlc_code(n, cn, len, code);
  if (!E) { b.pos += len; return len; }
  if (len > 32) { bs_put_t<E>(b, len - 32, 0); bs_put_t<E>(b, 32, code); }
  else bs_put_t<E>(b, len, code);
  return len;
}
TK_DEV int bs_vlc(BitSink& b, int n, uint32_t cn) { return bs_vlc_t<true>(b, n, cn); }
TK_DEV int vlc_len(int n, uint32_t cn) {
  int len; uint32_t code;
  vlc_code(n, cn, len, code);
  return len;
}

// write_mv (enc/write_bits.c:123-143): mvd via VLC 7 + sign bit per component, x first.
template <bool E> TK_DEV void bs_mv_t(BitSink& b, mv_t mv, mv_t mvp) {
  int dx = (int16_t)(mv.x - mvp.x), dy = (int16_t)(mv.y - mvp.y);
  uint32_t ax = (uint16_t)iabs(dx), ay = (uint16_t)iabs(dy);
  bs_vlc_t<E>(b, 7, ax);
  if (ax > 0) bs_put_t<E>(b, 1, dx < 0);
  bs_vlc_t<E>(b, 7, ay);
  if (ay > 0) bs_put_t<E>(b, 1, dy < 0);
}

// write_coeff (enc/write_bits.c:145-241). coeff: qsize x qsize row-major (qsize=min(size,16)),
// type bit0 = chroma, bit1 = intra block.  get(pos) -> coefficient at scan position pos; last_pos: the last non-zero scan position (0 if none).
template <class GetF> TK_DEV void bs_coeff_body(BitSink& b, GetF get, int N, int last_pos, int size, int type) {
  const int chroma = type & 1, intra = (type >> 1) & 1;
  int vlc_adaptive = intra && !chroma;
  const uint32_t eob_pos = chroma ? 0u : 2u;
  const int runtab = (chroma && size <= 8) ? 10 : 6;
  int pos = 0;
  if (chroma) {
    int c0 = get(0);
    if (last_pos == 0 && iabs(c0) == 1) { bs_put(b, 2, 2u + (c0 < 0)); pos = N; }
    else bs_put(b, 1, 0);
  }
  int level_mode = 1, level = 1;
  while (pos <= last_pos) {
    int c;
    if (level_mode) {
      while (pos <= last_pos && level > 0) {
        c = get(pos++);
        level = iabs(c);
        bs_vlc(b, vlc_adaptive, (uint32_t)level);
        if (level > 0) bs_put(b, 1, c < 0);
        if (!chroma) vlc_adaptive = level > 3;
      }
    }
    int run = 0;
    c = 0;
    while (c == 0 && pos <= last_pos) {
      c = get(pos++);
      run += !c;
      if (c) {
        level = iabs(c);
        int sign = c < 0;
        uint32_t cn = (level == 1) ? (uint32_t)((run * 5) / 4) : (uint32_t)(run * 5 + 4);
        bs_vlc(b, runtab, cn + (cn >= eob_pos));
        level_mode = level > 1;
        if (level > 1) bs_vlc(b, 0, (uint32_t)((level - 2) * 2 + sign));
        else bs_put(b, 1, sign);
        run = 0;
      }
    }
  }
  if (pos < N && level_mode) { bs_vlc(b, vlc_adaptive, 0); pos++; }
  if (pos < N) bs_vlc(b, runtab, eob_pos);
}
// one lane alone
TK_DEVNI void bs_coeff(BitSink& b, const int16_t* coeff, int size, int type) {
  const int qsize = size < kMaxQuant ? size : kMaxQuant;
  const int N = qsize * qsize;
  const int16_t* izz = qsize == 4 ? TK_TAB.izz4 : (qsize == 8 ? TK_TAB.izz8 : TK_TAB.izz16);
  int last_pos = N - 1;
  while (last_pos > 0 && coeff[izz[last_pos]] == 0) last_pos--;
  bs_coeff_body(b, [&](int pos) -> int { return (int)coeff[izz[pos]]; }, N, last_pos, size, type);
}
// Every lane of the team (cooperative emission, b.store marks the lane that writes).  The one-lane form walks the scan with two
// dependent memory round trips per position (scan table, coefficient); here the lanes fetch the coefficients of all scan positions
// at once (up to 256 = four per lane), the last non-zero position comes out of a ballot, and the sequential run / level automaton
// runs on wave-uniform values read back with v_readlane - scalar code, no memory access per position.
TK_DEVNI void bs_coeff_team(BitSink& b, const Team t, const int16_t* coeff, int size, int type) {
  const int qsize = size < kMaxQuant ? size : kMaxQuant;
  const int N = qsize * qsize;
#if TK_HOST
  (void)t;
  const int16_t* izz = qsize == 4 ? TK_TAB.izz4 : (qsize == 8 ? TK_TAB.izz8 : TK_TAB.izz16);
  int last_pos = N - 1;
  while (last_pos > 0 && coeff[izz[last_pos]] == 0) last_pos--;
  bs_coeff_body(b, [&](int pos) -> int { return (int)coeff[izz[pos]]; }, N, last_pos, size, type);
#else
  coeff = tk_uniform_ptr(coeff);
  const IzzRef izzr = izz_ref(t, qsize);
  int v[4];
  int last_pos = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int p = q * 64 + t.rank;
    v[q] = p < N ? (int)coeff[izzr.z[p]] : 0;
    const unsigned long long m = __ballot(v[q] != 0);
    if (m) last_pos = q * 64 + top_set(m);
  }
  last_pos = tk_uniform(last_pos);
  bs_coeff_body(b, [&](int pos) -> int {
    pos = tk_uniform(pos);
    const int q = pos >> 6;
    const int x = q == 0 ? v[0] : q == 1 ? v[1] : q == 2 ? v[2] : v[3];
    return __builtin_amdgcn_readlane(x, pos & 63);
  }, N, last_pos, size, type);
#endif
}

// Everything write_block / write_super_mode read besides the block parameters.
struct SynCtx {
  int frame_type, num_ref, enable_bipred, interp_ref;
  int max_pb_part, max_tb_part, num_intra_modes;
  int size, encode_this_size;
  int ctx_index, ctx_cbp;       // block_context_t index / cbp (common_block.c:283-309)
  int num_skip, num_merge;
  mv_t mvp;
};

struct BlkParam {  // block_param_t without the coefficient arrays (common/types.h:205-222)
  int8_t mode, intra_mode, skip_idx, pb_part, ref0, ref1, dir, tb_param, tb_split;
  uint8_t cbp_y, cbp_u, cbp_v;  // bit masks (4 TUs, MSB = TU0) when tb_split, else 0/1
  mv_t mv0[4], mv1[4];
};

// Number of bits write_coeff produces, evaluated W = team.size scan positions at a time (count only;
// must be called by ALL lanes of the team).  The level/run mode is again a {identity, ->run, ->level}
// automaton (zero -> run mode, |c| > 1 -> level mode, |c| == 1 keeps the mode), so ballots give each
// position its mode, its adaptive-VLC flag and its run length.  With W = 1 this is the serial loop.
// SC: address space of the coefficient block (SP_LDS for every buffer except the chroma buffers of tb-split 64/128 blocks)
template <int SC>
TK_DEVNI int coeff_bits_team(const Team t, const int16_t* coeff_, int size, int type) {
  coeff_ = tk_uniform_ptr(coeff_); size = tk_uniform(size); type = tk_uniform(type);
  const auto coeff = spc<SC>(coeff_);
  const int qsize = size < kMaxQuant ? size : kMaxQuant;
  const int N = qsize * qsize;
  const IzzRef izzr = izz_ref(t, qsize);
  const int chroma = type & 1, intra = (type >> 1) & 1;
  const uint32_t eob_pos = chroma ? 0u : 2u;
  const int runtab = (chroma && size <= 8) ? 10 : 6;
  const int W = t.size;
  int last = 0;
  for (int base = 0; base < N; base += W) {
    const int p = base + t.rank;
    const unsigned long long m = team_ballot(t, p < N && coeff[izzr.z[p]] != 0);
    if (m) last = base + top_set(m);
  }
  int bits = 0;  // per-lane partial, reduced at the end
  int head = 0;  // uniform part
  if (chroma) {
    const int c0 = coeff[izzr.z[0]];
    if (last == 0 && iabs(c0) == 1) return 2;
    head = 1;
  }
  int carryL = 1;                        // mode entering the round (1 = level mode)
  int carry_prevL = 0, carry_gt3 = 0;    // state of position base-1
  int carry_q = -1;
  int lastL = 0, last_gt3 = 0;
  for (int base = 0; base <= last; base += W) {
    const int p = base + t.rank;
    const int active = p <= last;
    const int c = active ? (int)coeff[izzr.z[p]] : 0;
    const int a = iabs(c);
    const unsigned long long mK = team_ballot(t, active && a != 1), mBig = team_ballot(t, active && a > 1);
    const int j = prev_set(mK, t.rank);
    const int modeL = j < 0 ? carryL : (int)((mBig >> j) & 1ull);
    const unsigned long long mL = team_ballot(t, active && modeL), mGt3 = team_ballot(t, active && a > 3);
    const unsigned long long mQ = team_ballot(t, active && (a != 0 || modeL));
    if (active) {
      if (modeL) {
        int adaptive = 0;
        if (!chroma) {
          if (p == 0) adaptive = intra;
          else {
            const int prevL = t.rank > 0 ? (int)((mL >> (t.rank - 1)) & 1ull) : carry_prevL;
            const int prevG = t.rank > 0 ? (int)((mGt3 >> (t.rank - 1)) & 1ull) : carry_gt3;
            adaptive = prevL ? prevG : 0;
          }
        }
        bits += vlc_len(adaptive, (uint32_t)a) + (a > 0);
      } else if (a != 0) {
        const int jq = prev_set(mQ, t.rank);
        const int q = jq < 0 ? carry_q : base + jq;
        const int run = p - 1 - q;
        const uint32_t cn = (a == 1) ? (uint32_t)((run * 5) / 4) : (uint32_t)(run * 5 + 4);
        bits += vlc_len(runtab, cn + (cn >= eob_pos));
        bits += (a > 1) ? vlc_len(0, (uint32_t)((a - 2) * 2 + (c < 0))) : 1;
      }
    }
    // state of the coefficient at `last` (for the trailing symbols)
    if (last >= base && last < base + W) {
      const int li = last - base;
      lastL = (int)((mL >> li) & 1ull);
      last_gt3 = (int)((mGt3 >> li) & 1ull);
    }
    const int jj = top_set(mK);
    if (jj >= 0) carryL = (int)((mBig >> jj) & 1ull);
    const int wl = W - 1;
    carry_prevL = (int)((mL >> wl) & 1ull);
    carry_gt3 = (int)((mGt3 >> wl) & 1ull);
    const int jq2 = top_set(mQ);
    if (jq2 >= 0) carry_q = base + jq2;
  }
  int pos = last + 1;
  // carryL is now the mode after consuming coefficient `last`
  if (pos < N && carryL) {
    const int adaptive = chroma ? 0 : (lastL ? last_gt3 : 0);
    head += vlc_len(adaptive, 0);
    pos++;
  }
  if (pos < N) head += vlc_len(runtab, eob_pos);
  return head + team_sum(t, bits);
}

// write_super_mode (enc/write_bits.c:257-358).
template <bool E> TK_DEV void bs_super_mode_t(BitSink& b, const SynCtx& s, int mode, int ref0, int split_flag) {
  if (s.frame_type != F_I) {
    if (!s.encode_this_size) { bs_put_t<E>(b, 1, !split_flag); return; }
    int bipred_possible = s.num_ref > 1 && s.enable_bipred;
    int split_possible = s.size > kMinBlk;
    int maxbit = 2 + s.num_ref + split_possible + bipred_possible;
    if (s.interp_ref > 2) maxbit -= 1;
    int moved = (s.ctx_index == 2 || s.ctx_index > 3);
    if (split_flag == 1) {
      if (s.size > 128) { bs_put_t<E>(b, 1, 0); return; }
      int code = 1;
      if (moved) code = (code + 3) % 4;
      bs_vlc_t<E>(b, 10 + maxbit, (uint32_t)code);
      return;
    }
    int code = 0;
    if (s.interp_ref) {
      if (mode == M_SKIP) code = 0;
      else if (mode == M_MERGE) code = 2;
      else if (mode == M_BIPRED) code = 3;
      else if (mode == M_INTRA) code = 4;
      else if (mode == M_INTER && ref0 > 0) code = 4 + ref0;
      else code = 4 + s.num_ref;
      if (!bipred_possible && code > 3) code--;
      if (!split_possible && code > 1) code--;
      if (moved && s.size > kMinBlk && code < 3) code = (code + 2) % 3;
    } else {
      if (mode == M_SKIP) code = 0;
      else if (mode == M_INTER && ref0 == 0) code = 2;
      else if (mode == M_MERGE) code = 3;
      else if (mode == M_BIPRED) code = 4;
      else if (mode == M_INTRA) code = 5;
      else if (mode == M_INTER && ref0 > 0) code = 5 + ref0;
      if (!bipred_possible && code > 4) code--;
      if (!split_possible && code > 1) code--;
      if (moved && s.size > kMinBlk && code < 4) code = (code + 3) % 4;
    }
    bs_vlc_t<E>(b, 10 + maxbit, (uint32_t)code);
  } else {
    if (s.encode_this_size && (s.size > kMinBlk || split_flag == 1)) bs_put_t<E>(b, 1, (uint32_t)split_flag);
  }
}

TK_DEV void bs_super_mode(BitSink& b, const SynCtx& s, int mode, int ref0, int split_flag) { bs_super_mode_t<true>(b, s, mode, ref0, split_flag); }

TK_DEV int cbp_code(int cbp) {  // cbp_table (enc/write_bits.c:382)
  return cbp == 0 ? 1 : cbp == 1 ? 0 : cbp == 2 ? 5 : cbp == 3 ? 2 : cbp == 4 ? 6 : cbp == 5 ? 3 : cbp == 6 ? 7 : 4;
}

// write_block (enc/write_bits.c:360-600).  cy/cu/cv: quantised coefficients, TU t of a
// tb-split block at offset t*256 (MAX_QUANT_SIZE^2) like the reference.
// `t`: team for cooperative counting (b.emit == 0, all lanes call) or nullptr (single-lane emission).
template <bool E, int SC> TK_DEV void bs_coeff_any(BitSink& b, const Team* t, const int16_t* coeff, int size, int type) {
  if (!E) b.pos += coeff_bits_team<SC>(*t, coeff, size, type);
  else if (t) bs_coeff_team(b, *t, coeff, size, type);   // cooperative emission
  else bs_coeff(b, coeff, size, type);
}

// ybits (optional, counting mode only): bit lengths of the luma TU coefficient strings already counted by the
// caller (partial-cost pruning), [0] for an unsplit block, [t] for TU t of a tb-split one.
TK_DEV mv_t uniform_mv(mv_t m) { return mk_mv(tk_uniform(m.x), tk_uniform(m.y)); }
TK_DEV SynCtx uniform_syn(const SynCtx& a) {
  SynCtx u;
  u.frame_type = tk_uniform(a.frame_type); u.num_ref = tk_uniform(a.num_ref); u.enable_bipred = tk_uniform(a.enable_bipred);
  u.interp_ref = tk_uniform(a.interp_ref); u.max_pb_part = tk_uniform(a.max_pb_part); u.max_tb_part = tk_uniform(a.max_tb_part);
  u.num_intra_modes = tk_uniform(a.num_intra_modes); u.size = tk_uniform(a.size); u.encode_this_size = tk_uniform(a.encode_this_size);
  u.ctx_index = tk_uniform(a.ctx_index); u.ctx_cbp = tk_uniform(a.ctx_cbp); u.num_skip = tk_uniform(a.num_skip);
  u.num_merge = tk_uniform(a.num_merge); u.mvp = uniform_mv(a.mvp);
  return u;
}
TK_DEV BlkParam uniform_blk(const BlkParam& a) {
  BlkParam u;
  u.mode = (int8_t)tk_uniform(a.mode); u.intra_mode = (int8_t)tk_uniform(a.intra_mode); u.skip_idx = (int8_t)tk_uniform(a.skip_idx);
  u.pb_part = (int8_t)tk_uniform(a.pb_part); u.ref0 = (int8_t)tk_uniform(a.ref0); u.ref1 = (int8_t)tk_uniform(a.ref1);
  u.dir = (int8_t)tk_uniform(a.dir); u.tb_param = (int8_t)tk_uniform(a.tb_param); u.tb_split = (int8_t)tk_uniform(a.tb_split);
  u.cbp_y = (uint8_t)tk_uniform(a.cbp_y); u.cbp_u = (uint8_t)tk_uniform(a.cbp_u); u.cbp_v = (uint8_t)tk_uniform(a.cbp_v);
  for (int i = 0; i < 4; i++) { u.mv0[i] = uniform_mv(a.mv0[i]); u.mv1[i] = uniform_mv(a.mv1[i]); }
  return u;
}

// The part of write_block that does not depend on the residual: super-mode, intra mode / partition and vector differences /
// candidate index (enc/write_bits.c:360-470).  bs_block_t starts with it; the RDO trials use its length as the first term of
// their lower bounds (tk_block.h: PruneCtx::head_bits).
template <bool E> TK_DEV void bs_block_head_t(BitSink& b, const SynCtx& s, const BlkParam& p) {
  const int mode = p.mode;
  bs_super_mode_t<E>(b, s, mode, p.ref0, 0);
  if (mode == M_INTRA) {
    if (s.num_intra_modes <= 4) bs_put_t<E>(b, 2, (uint32_t)p.intra_mode);
    else bs_vlc_t<E>(b, 8, (uint32_t)p.intra_mode);
  } else if (mode == M_INTER) {
    if (s.max_pb_part > 1) bs_vlc_t<E>(b, 13, (uint32_t)p.pb_part);
    mv_t mvp2 = s.mvp;
    bs_mv_t<E>(b, p.mv0[0], mvp2);
    mvp2 = p.mv0[0];
    if (p.pb_part == P_HOR) bs_mv_t<E>(b, p.mv0[2], mvp2);
    else if (p.pb_part == P_VER) bs_mv_t<E>(b, p.mv0[1], mvp2);
    else if (p.pb_part == P_QUAD) { bs_mv_t<E>(b, p.mv0[1], mvp2); bs_mv_t<E>(b, p.mv0[2], mvp2); bs_mv_t<E>(b, p.mv0[3], mvp2); }
  } else if (mode == M_BIPRED) {
    mv_t mvp2 = s.mvp;
    if (p.pb_part == P_NONE) bs_mv_t<E>(b, p.mv0[0], mvp2);
    if (s.frame_type == F_B) mvp2 = p.mv0[0];
    bs_mv_t<E>(b, p.mv1[0], mvp2);
    if (p.pb_part != P_NONE) {
      mvp2 = p.mv1[0];
      if (p.pb_part == P_HOR) bs_mv_t<E>(b, p.mv1[2], mvp2);
      else if (p.pb_part == P_VER) bs_mv_t<E>(b, p.mv1[1], mvp2);
      else { bs_mv_t<E>(b, p.mv1[1], mvp2); bs_mv_t<E>(b, p.mv1[2], mvp2); bs_mv_t<E>(b, p.mv1[3], mvp2); }
    }
    if (s.frame_type == F_P) {
      if (s.num_ref == 2) bs_vlc_t<E>(b, 13, (uint32_t)(2 * p.ref0 + p.ref1));
      else bs_vlc_t<E>(b, 10, (uint32_t)(4 * p.ref0 + p.ref1));
    }
  } else if (mode == M_SKIP || mode == M_MERGE) {
    int nvec = mode == M_SKIP ? s.num_skip : s.num_merge;
    if (nvec == 4) bs_put_t<E>(b, 2, (uint32_t)p.skip_idx);
    else if (nvec == 3) bs_vlc_t<E>(b, 12, (uint32_t)p.skip_idx);
    else if (nvec == 2) bs_put_t<E>(b, 1, (uint32_t)p.skip_idx);
  }
}

// E = false: cooperative counting (every lane of the team calls, b.emit == 0, tm != nullptr) - the instance all RDO trials
// use; E = true: emission - by every lane of the team (tm != nullptr, BitSink::store marks the lane that writes) or by one lane alone.
// SCC: address space of the CHROMA coefficient buffers in counting mode (luma is always SP_LDS on the device).
template <bool E, int SCC = SP_LDS>
TK_DEVNI int bs_block_t(BitSink& b, const SynCtx& s_in, const BlkParam& p_in, const int16_t* cy, const int16_t* cu,
                    const int16_t* cv, const Team* tm, const int* ybits) {
  // cooperative counting, or cooperative emission (tm != nullptr: every lane of the team runs the emission on the same values);
  // otherwise the emitting call is made by one lane alone
  const bool coop = !E || tm != nullptr;
  const SynCtx s = coop ? uniform_syn(s_in) : s_in;
  const BlkParam p = coop ? uniform_blk(p_in) : p_in;
  const int start = b.pos;
  const int size = s.size, size_uv = size >> 1;
  const int mode = p.mode;
  const int coeff_type = (mode == M_INTRA) << 1;
  // coefficient offset of TU t of a tb-split block: t * qs^2, qs = min(TU size, 16)
  const int qy = size / 2 < kMaxQuant ? size / 2 : kMaxQuant, qc = size_uv / 2 < kMaxQuant ? size_uv / 2 : kMaxQuant;
  const int sty = qy * qy, stc = qc * qc;
  bs_block_head_t<E>(b, s, p);

  if (mode != M_SKIP) {
    const int tb_split = p.tb_split;
    int code;
    const int off = mode == M_MERGE ? 1 : 2;
    if (s.max_tb_part > 1 && tb_split) {
      code = off;
    } else {
      int cbp = p.cbp_y + (p.cbp_u << 1) + (p.cbp_v << 2);
      code = cbp_code(cbp);
      if (mode == M_MERGE) {
        if (code == 1) code = 7;
        else if (code > 1) code--;
      } else if (s.ctx_cbp == 0 && code < 2) code = 1 - code;
      if (s.max_tb_part > 1 && code >= off) code++;
    }
    bs_vlc_t<E>(b, 0, (uint32_t)code);

    if (tb_split == 0) {
      if (p.cbp_y) { if (!E && ybits) b.pos += ybits[0]; else bs_coeff_any<E, SP_LDS>(b, tm, cy, size, coeff_type | 0); }
      if (p.cbp_u) bs_coeff_any<E, SCC>(b, tm, cu, size_uv, coeff_type | 1);
      if (p.cbp_v) bs_coeff_any<E, SCC>(b, tm, cv, size_uv, coeff_type | 1);
    } else if (size_uv > 4) {
      for (int t = 0; t < 4; t++) {
        int ty = (p.cbp_y >> (3 - t)) & 1, tu = (p.cbp_u >> (3 - t)) & 1, tv = (p.cbp_v >> (3 - t)) & 1;
        int c = cbp_code(ty + (tu << 1) + (tv << 2));
        if (s.ctx_cbp == 0 && c < 2) c = 1 - c;
        bs_vlc_t<E>(b, 0, (uint32_t)c);
        if (ty) { if (!E && ybits) b.pos += ybits[t]; else bs_coeff_any<E, SP_LDS>(b, tm, cy + t * sty, size / 2, coeff_type | 0); }
        if (tu) bs_coeff_any<E, SCC>(b, tm, cu + t * stc, size_uv / 2, coeff_type | 1);
        if (tv) bs_coeff_any<E, SCC>(b, tm, cv + t * stc, size_uv / 2, coeff_type | 1);
      }
    } else {
      for (int t = 0; t < 4; t++) {
        int ty = (p.cbp_y >> (3 - t)) & 1;
        bs_put_t<E>(b, 1, (uint32_t)ty);
        if (ty) { if (!E && ybits) b.pos += ybits[t]; else bs_coeff_any<E, SP_LDS>(b, tm, cy + t * sty, size / 2, coeff_type | 0); }
      }
      bs_vlc_t<E>(b, 13, (uint32_t)(p.cbp_u + 2 * p.cbp_v));
      if (p.cbp_u) bs_coeff_any<E, SCC>(b, tm, cu, size_uv, coeff_type | 1);
      if (p.cbp_v) bs_coeff_any<E, SCC>(b, tm, cv, size_uv, coeff_type | 1);
    }
  }
  return b.pos - start;
}
}  // namespace tk
