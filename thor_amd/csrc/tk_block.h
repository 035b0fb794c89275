// tk_block.h - the per-superblock decision engine: quadtree recursion, early skip, RDO trials,
// final encode (recon + cell state + bit emission).
// Specification followed: enc/encode_block.c:2401-2566 (process_block), :1835-2121
// (mode_decision_rdo), :1340-1514 (encode_block), :1100-1338 (encode_and_reconstruct_*),
// :1679-1833 (search_bipred_prediction_params), :1033-1098 (search_inter_prediction_params),
// :2123-2392 (early skip), :916-926 (cost_calc), :1568-1613 (copy_deblock_data);
// common/inter_prediction.c:413-834 (get_mv_pred / get_mv_merge / get_mv_skip);
// common/common_block.c:283-309 (find_block_contexts), :347-428 (improve_uv_prediction);
// common/common_block.h:52-95 (availability).
#pragma once
#include "tk_common.h"
#include "tk_bits.h"
#include "tk_pred.h"
#include "tk_xform.h"
#include "tk_me.h"


namespace tk {

struct Node {
  int size, ypos, xpos, bw, bh;
  int stage, child;
  int md_done;          // top-down flow (encoder_speed > 0): mode decision already made, cost in cost_this
  unsigned cost_this;
  unsigned cost_small;
  int bitpos0;
  int encode_this_size, encode_rect;
  SynCtx syn;
  InterPred skip[2], merge[2];
  BlkParam best;
};

// Working set.  Per workgroup (= per superblock in flight): WgShared - constant tables, the per-SB candidate
// lists, the recursion stack of the master wave and the fork/join state of the parallel block decision.  Per
// wavefront: SmallWs (LDS: transform tiles, ME scratch, coefficient buffers, intra edges) and BigWs (sample blocks
// up to 128x128 in a global scratch arena that stays L1/L2 resident).
enum { kProfSlots = 32 };
enum { kMdMaxItems = 48 };
#ifndef TK_LDSBLK
#define TK_LDSBLK 16
#endif
enum { kLdsBlk = TK_LDSBLK };   // coding blocks up to this size keep their sample buffers in LDS (16 or 32)
enum { MD_SKIP = 0, MD_MERGE, MD_REF, MD_INTRA, MD_BIPRED, MD_TRIAL, MD_BIJOINT };
enum { WG_CMD_EXIT = 0, WG_CMD_MD = 1 };
struct MdItem { int8_t kind, a, b, pad; };
struct WgShared {
  XformTabs tabs;
  MeLists lists;
  Node stack[5];
  // ---- parallel block decision (mode_decision_par)
  int cmd;                       // what the parked waves do after the next workgroup barrier
  int next_item, n_items;        // work queue cursor (atomic) / length
  int refs_done, n_ref_items;    // reference searches finished (atomic) / expected; the last one triggers the bipred item
  int do_bipred;                 // 0 none, 1 one item (B frames), 2 lock-step phase after the queue (P frames)
  // A reference's MD_REF item searches its partitions one after the other and publishes each partition's vectors; the RDO
  // trials of (reference, partition) are queue items of their own (MD_TRIAL) that any wave takes once the vectors are there.
  int parts_done[kMaxRefs];      // partitions of reference r searched so far (atomic, released after ref_mv[r][part] is written)
  mv_t ref_mv[kMaxRefs][4][4];   // [reference][partition][quadrant]
  // bi-prediction search of P frames, run by all waves in lock step (bipred_par)
  const void* bp_org8;           // 2*org - pred of the current step (leader's buffer)
  unsigned bp_sad2[2][kMaxRefs];   // results of a lock-step search step, double-buffered (bipred_par)
  mv_t bp_mv2[2][kMaxRefs][4];
  // B frames: the telescope of the joint +mv / -mv search (motion_estimate_bi) runs as a queue item of its own (MD_BIJOINT) as soon as
  // the PART_NONE vector of its first reference is known; whoever gets there first - the item or the bi-prediction item that
  // needs its result - claims it (0 -> 1) and publishes the result (-> 2)
  int bj_state;
  unsigned bj_sad;
  mv_t bj_mv;
  int node;                      // index of the node being decided in `stack`
  mv_t mvp;
  mv_t mv_center[kMaxRefs];
  unsigned long long bestkey;    // min over finished trials of (cost << 32 | evaluation order), atomic
  unsigned long long wkey[kWaves];
  BlkParam wbest[kWaves];
  void* wsnap[kWaves];           // BigWs of wave w (its best-trial snapshot)
  MdItem items[kMdMaxItems];
  // original samples (Y, U, V; stride = block size) of the coding block being decided when it is at most kLdsBlk wide:
  // loaded once by the master, read by every trial of every wave instead of the frame in global memory
  alignas(16) unsigned char org_raw[kLdsBlk * kLdsBlk * 3];
};
// Bytes of per-wave LDS that extend the motion search's window beyond the transform workspace it borrows (tk_me.h:MeWin).  The
// 8-bit kernel runs three workgroups per CU (53 KB of LDS each; two measured 12 % slower at full load, profiles/r04_call2_ab.md):
// 944 bytes more take 32x32 PUs with a reach of 16 samples.  The 16-bit kernel needs 224 VGPRs and runs two workgroups per CU
// anyway (80 KB each): 16-bit PUs up to 32x32 with a reach of 16.  -DTK_OCC=2 builds: 8-bit PUs up to 64x64.
template <typename PIX> struct WinExtra { enum { bytes = sizeof(PIX) == 1 ? (TK_OCC == 2 ? 7680 : 944) : 5376 }; };
template <typename PIX> struct SmallWs {
  XformWs xf;
  alignas(16) unsigned char win_extra[WinExtra<PIX>::bytes];   // must directly follow xf
  MeWs me;
  IntraEdge<PIX> edge;
  // quantised coefficients of the current trial; TU t of a tb-split block at offset t * qs^2 with
  // qs = min(TU size, 16).  Chroma needs more than 256 entries only for tb-split 64/128 blocks, which
  // use the BigWs buffers instead.
  int16_t coef_y[4 * 256], coef_u[256], coef_v[256];
  unsigned long long acc[12];
  // sample blocks (prediction, the two bi-prediction inputs, reconstruction, 2*org-pred) of coding blocks up to
  // kLdsBlk x kLdsBlk: the trials of the small blocks - the bulk of all trials - never round-trip through global memory
  alignas(16) PIX lbuf[7 * kLdsBlk * kLdsBlk];
#if defined(THOR_PROF)
  long long prof[kProfSlots];
#else
  long long prof[1];
#endif
};
// the row-segment loads of the motion search read the LDS sample blocks with 16-byte ds_read
static_assert(offsetof(SmallWs<uint8_t>, lbuf) % 16 == 0 && offsetof(SmallWs<uint16_t>, lbuf) % 16 == 0 && sizeof(SmallWs<uint8_t>) % 16 == 0 &&
              sizeof(SmallWs<uint16_t>) % 16 == 0 && offsetof(WgShared, org_raw) % 16 == 0, "LDS sample blocks must be 16-byte aligned");
template <typename PIX> struct BigWs {
  PIX pred_y[kMaxSb * kMaxSb], pred_u[kMaxSb * kMaxSb / 4], pred_v[kMaxSb * kMaxSb / 4];
  PIX p0_y[kMaxSb * kMaxSb], p0_u[kMaxSb * kMaxSb / 4], p0_v[kMaxSb * kMaxSb / 4];
  PIX p1_y[kMaxSb * kMaxSb], p1_u[kMaxSb * kMaxSb / 4], p1_v[kMaxSb * kMaxSb / 4];
  PIX rec_y[kMaxSb * kMaxSb], rec_u[kMaxSb * kMaxSb / 4], rec_v[kMaxSb * kMaxSb / 4];
  PIX org8[kMaxSb * kMaxSb];
  int16_t coef_u_big[4 * 256], coef_v_big[4 * 256];
  // snapshot of this wave's best trial of the current block decision (reconstruction + quantised coefficients): the final
  // encode of the winning trial copies it instead of predicting / transforming the block again (mode_decision_par)
  PIX best_y[kMaxSb * kMaxSb], best_u[kMaxSb * kMaxSb / 4], best_v[kMaxSb * kMaxSb / 4];
  int16_t best_cy[4 * 256], best_cu[4 * 256], best_cv[4 * 256];
};
template <typename PIX> struct TeamWs {  // view (lives in registers)
  XformWs* xfp;
  MeWs* mep;
  IntraEdge<PIX>* edgep;
  int16_t *coef_y, *coef_u, *coef_v;          // current (may point at the big chroma buffers)
  int16_t *coef_u_small, *coef_v_small, *coef_u_big, *coef_v_big;
  unsigned long long* acc;
  WgShared* sh;
  Node* stack;
  long long* prof;
  PIX *pred_y, *pred_u, *pred_v, *p0_y, *p0_u, *p0_v, *p1_y, *p1_u, *p1_v, *rec_y, *rec_u, *rec_v, *org8;  // current (ws_select)
  BigWs<PIX>* big;
  PIX* lbuf;
  const PIX *org_y, *org_u, *org_v;  // original samples of the current coding block (origin), strides org_sy / org_sc
  int org_sy, org_sc;
};
// On the device the per-wave view and the frame job live in LDS and are passed around as LDS-typed pointer / reference
// (ds_read of the members instead of generic loads); plain pointer / reference on the host.
#if TK_HOST
template <typename PIX> using WsP = TeamWs<PIX>*;
template <typename PIX> using JobR = const FrameJob<PIX>&;
#else
template <typename PIX> using WsP = TK_LDS TeamWs<PIX>*;
template <typename PIX> using JobR = const TK_LDS FrameJob<PIX>&;
#endif
// Point the sample-block views at the LDS buffers (coding blocks up to kLdsBlk) or at the global scratch slot.
template <class WP> TK_DEV void ws_select(WP w, int size) {  // WP: TeamWs<PIX>* in any address space
  if (size <= kLdsBlk) {
    auto b = w->lbuf;
    const int n = size * size, c = n >> 2;
    w->pred_y = b; w->pred_u = b + n; w->pred_v = b + n + c; b += n + 2 * c;
    w->p0_y = b; w->p0_u = b + n; w->p0_v = b + n + c; b += n + 2 * c;
    w->p1_y = b; w->p1_u = b + n; w->p1_v = b + n + c; b += n + 2 * c;
    w->rec_y = b; w->rec_u = b + n; w->rec_v = b + n + c; b += n + 2 * c;
    w->org8 = b;
  } else {
    auto g = w->big;
    w->pred_y = g->pred_y; w->pred_u = g->pred_u; w->pred_v = g->pred_v;
    w->p0_y = g->p0_y; w->p0_u = g->p0_u; w->p0_v = g->p0_v;
    w->p1_y = g->p1_y; w->p1_u = g->p1_u; w->p1_v = g->p1_v;
    w->rec_y = g->rec_y; w->rec_u = g->rec_u; w->rec_v = g->rec_v; w->org8 = g->org8;
  }
}
// Point ws->org_* at the original samples of coding block `nd`: the frame planes, or (blocks up to kLdsBlk) the
// workgroup's LDS copy, which the master fills with load = 1 before any wave uses it.
template <typename PIX>
TK_DEV void org_select(const Team t, JobR<PIX> J, WsP<PIX> w, int size, int ypos, int xpos, int bw, int bh, int load) {
  if (size <= kLdsBlk) {
    PIX* b = (PIX*)w->sh->org_raw;
    const int n = size * size, sc = size >> 1;
    if (load) {
      t.sync();
      const Div2 dw = mk_div(bw), dc = mk_div(bw >> 1);
      const TK_GLOBAL PIX* gy = gptr(J.orig.y + ypos * J.orig.sy + xpos);
      const TK_GLOBAL PIX* gu = gptr(J.orig.u + (ypos >> 1) * J.orig.sc + (xpos >> 1));
      const TK_GLOBAL PIX* gv = gptr(J.orig.v + (ypos >> 1) * J.orig.sc + (xpos >> 1));
      const auto bl = ldsc(b);
      for (int k = t.rank; k < bw * bh; k += t.size) { int i, j; split2(dw, k, i, j); bl[i * size + j] = gy[i * J.orig.sy + j]; }
      for (int k = t.rank; k < (bw >> 1) * (bh >> 1); k += t.size) {
        int i, j;
        split2(dc, k, i, j);
        bl[n + i * sc + j] = gu[i * J.orig.sc + j];
        bl[n + (n >> 2) + i * sc + j] = gv[i * J.orig.sc + j];
      }
      t.sync();
    }
    w->org_y = b; w->org_u = b + n; w->org_v = b + n + (n >> 2);
    w->org_sy = size; w->org_sc = sc;
  } else {
    w->org_y = J.orig.y + ypos * J.orig.sy + xpos;
    w->org_u = J.orig.u + (ypos >> 1) * J.orig.sc + (xpos >> 1);
    w->org_v = J.orig.v + (ypos >> 1) * J.orig.sc + (xpos >> 1);
    w->org_sy = J.orig.sy; w->org_sc = J.orig.sc;
  }
}

template <typename PIX> TK_DEV TeamWs<PIX> make_ws(SmallWs<PIX>* s, WgShared* sh, BigWs<PIX>* g) {
  TeamWs<PIX> w;
  w.xfp = &s->xf; w.mep = &s->me; w.edgep = &s->edge;
  w.sh = sh; s->xf.tabs = &sh->tabs; s->me.lists = &sh->lists;
  // the search window of a motion search lives in the transform workspace (in | tmp | coef: contiguous), idle during a search,
  // and continues into win_extra
  static_assert(offsetof(SmallWs<PIX>, win_extra) == offsetof(SmallWs<PIX>, xf) + sizeof(XformWs), "win_extra must directly follow the transform workspace");
  s->me.win = (uint32_t*)s->xf.in;
  s->me.win_cap = (int)(sizeof(XformWs) - offsetof(XformWs, in)) + (int)WinExtra<PIX>::bytes;
  s->me.cwin_valid = 0;
  w.coef_y = s->coef_y; w.coef_u = s->coef_u; w.coef_v = s->coef_v;
  w.coef_u_small = s->coef_u; w.coef_v_small = s->coef_v; w.coef_u_big = g->coef_u_big; w.coef_v_big = g->coef_v_big;
  w.acc = s->acc; w.stack = sh->stack; w.prof = s->prof;
  s->xf.prof = s->prof; s->me.prof = s->prof;
  w.big = g; w.lbuf = s->lbuf;
  w.org_y = w.org_u = w.org_v = nullptr; w.org_sy = w.org_sc = 0;
  ws_select(&w, kMaxSb);
  return w;
}

// -DTHOR_PROF -DTHOR_PROF_MD: slots 16..25 hold the time of the decision's work-queue items by kind (all waves) and of the phases
// the master runs alone, instead of the transform-unit sizes: 16 skip/merge items, 17 intra items, 18 search items (MD_REF),
// 19 trial items incl. their wait for the vectors, 20 wait of the trial items alone, 21 queue set-up (master), 22 block entry
// (contexts, candidates, original block), 23 early-skip path (check + trial + final encode), 24 final encode of decided blocks: bit emission
// (one lane), 25 final encode of decided blocks: reconstruction copy + cell state.
// -DTHOR_PROF_MD_PARTS=mask (default 7) keeps only some of them - 1: the items inside md_worker_sp's loop, 2: the trial items' wait, 4: the
// master's phases - and mask bit 8 makes the loop counters accumulate in registers and store once after the loop (bisection of the
// hang of the fully instrumented build, profiles/r04_call2_ab.md).
#if defined(THOR_PROF_MD) && defined(THOR_PROF) && !TK_HOST
#ifndef THOR_PROF_MD_PARTS
#define THOR_PROF_MD_PARTS 7
#endif
#define TK_PROFMD_MARK(v) TK_PROF_MARK(v)
#define TK_PROFMD_ACC(ws, id, v) TK_PROF_ACC(ws, id, v)
#define TK_PROFMD_CNT(ws, id) TK_PROF_CNT(ws, id)
#define TK_PROFMD_ON(bit) ((THOR_PROF_MD_PARTS) & (bit))
#else
#define TK_PROFMD_ON(bit) 0
#define TK_PROFMD_MARK(v) do {} while (0)
#define TK_PROFMD_ACC(ws, id, v) do {} while (0)
#define TK_PROFMD_CNT(ws, id) do {} while (0)
#endif
enum { PF_SB = 0, PF_ESKIP, PF_ME_FULL, PF_ME_SUB, PF_PRED_INTER, PF_PRED_INTRA, PF_TU, PF_BITS, PF_COST, PF_FINAL,
       PF_CFL, PF_BIPRED_PREP, PF_QUANT };

// ---------------------------------------------------------------------------------
// availability (common_block.h:52-95)
// ---------------------------------------------------------------------------------
TK_DEV int upright_avail(int ypos, int xpos, int bw, int bh, int fw, int sb) {
  int a = (ypos > 0) && (xpos + bw < fw);
  int size = bw > bh ? bw : bh;
  for (int s2 = size; s2 < sb; s2 *= 2)
    if ((ypos % (s2 << 1)) == s2 && (xpos % s2) == (s2 - size)) a = 0;
  return a;
}
TK_DEV int downleft_avail(int ypos, int xpos, int bw, int bh, int fh, int sb) {
  int a = (xpos > 0) && (ypos + bh < fh);
  int size = bw > bh ? bw : bh;
  if ((ypos % sb) == (sb - size) && (xpos % sb) == 0) a = 0;
  for (int s2 = 2 * size; s2 <= sb; s2 *= 2)
    if ((ypos % s2) == (s2 - size) && (xpos % s2) > 0) a = 0;
  return a;
}

TK_DEV InterPred zero_pred() {
  InterPred z;
  z.mv0 = mk_mv(0, 0);
  z.mv1 = mk_mv(0, 0);
  z.ref0 = z.ref1 = 0;
  z.dir = 0;
  z.pad = 0;
  return z;
}
TK_DEV InterPred cell_pred(const DbCell& c) {
  InterPred p;
  p.mv0 = c.mv0;
  p.mv1 = c.mv1;
  p.ref0 = c.ref0;
  p.ref1 = c.ref1;
  p.dir = c.dir;
  p.pad = 0;
  return p;
}

// get_mv_pred (inter_prediction.c:413-526): median of three neighbours' mv0.
TK_DEV mv_t get_mv_pred(const DbCell* cells, int cs, int ypos, int xpos, int fw, int fh, int size, int sb) {
  const int bsz = size / kMinPb;
  const int bi = (ypos / kMinPb) * cs + xpos / kMinPb;
  const int up0 = bi - cs, up1 = bi - cs + (bsz - 1) / 2, up2 = bi - cs + bsz - 1;
  const int l0 = bi - 1, l1 = bi + cs * ((bsz - 1) / 2) - 1, l2 = bi + cs * (bsz - 1) - 1;
  const int dl = bi + cs * bsz - 1, ur = bi - cs + bsz, ul = bi - cs - 1;
  const int U = ypos > 0, L = xpos > 0;
  const int UR = upright_avail(ypos, xpos, size, size, fw, sb);
  const int DL = downleft_avail(ypos, xpos, size, size, fh, sb);
  mv_t a = mk_mv(0, 0), b = a, c = a;
  if (U == 0 && UR == 0 && L == 0 && DL == 0) {
  } else if (U == 1 && UR == 0 && L == 0 && DL == 0) { a = cells[up0].mv0; b = cells[up1].mv0; c = cells[up2].mv0; }
  else if (U == 1 && UR == 1 && L == 0 && DL == 0) { a = cells[up0].mv0; b = cells[up2].mv0; c = cells[ur].mv0; }
  else if (U == 0 && UR == 0 && L == 1 && DL == 0) { a = cells[l0].mv0; b = cells[l1].mv0; c = cells[l2].mv0; }
  else if (U == 1 && UR == 0 && L == 1 && DL == 0) { a = cells[ul].mv0; b = cells[up2].mv0; c = cells[l2].mv0; }
  else if (U == 1 && UR == 1 && L == 1 && DL == 0) { a = cells[up0].mv0; b = cells[ur].mv0; c = cells[l2].mv0; }
  else if (U == 0 && UR == 0 && L == 1 && DL == 1) { a = cells[l0].mv0; b = cells[l2].mv0; c = cells[dl].mv0; }
  else if (U == 1 && UR == 0 && L == 1 && DL == 1) { a = cells[up2].mv0; b = cells[l0].mv0; c = cells[dl].mv0; }
  else if (U == 1 && UR == 1 && L == 1 && DL == 1) { a = cells[up0].mv0; b = cells[ur].mv0; c = cells[l0].mv0; }
  mv_t p;
  p.x = a.x < b.x ? tmin(b.x, tmax(a.x, c.x)) : tmin(a.x, tmax(b.x, c.x));
  p.y = a.y < b.y ? tmin(b.y, tmax(a.y, c.y)) : tmin(a.y, tmax(b.y, c.y));
  return p;
}

// get_mv_skip / get_mv_merge (LIMITED_SKIP variant; inter_prediction.c:528-834): identical rules.
TK_DEV int get_mv_cands(const DbCell* cells, int cs, int ypos, int xpos, int fw, int fh, int size, int sb,
                        InterPred* out) {
  const int bsz = size / kMinPb;
  const int bi = (ypos / kMinPb) * cs + xpos / kMinPb;
  int up0 = bi - cs, up2 = bi - cs + bsz - 1;
  int l0 = bi - 1, l2 = bi + cs * (bsz - 1) - 1;
  const int ur = bi - cs + bsz;
  const int U = ypos > 0, L = xpos > 0;
  const int UR = upright_avail(ypos, xpos, size, size, fw, sb);
  if (ypos + size > fh) l2 = l0;
  if (xpos + size > fw) up2 = up0;
  InterPred tmp[2];
  tmp[0] = L ? cell_pred(cells[l2]) : zero_pred();
  tmp[1] = UR ? cell_pred(cells[ur]) : (U ? cell_pred(cells[up2]) : zero_pred());
  out[0] = tmp[0];
  int n = 1;
  // duplicate test (inter_prediction.c:816-826); dir == -1 plays the reference's (uint32)-1
  const InterPred& q = tmp[1];
  const InterPred& o = out[0];
  int dup = q.mv0.x == o.mv0.x && q.mv0.y == o.mv0.y && q.ref0 == o.ref0 && q.mv1.x == o.mv1.x &&
            q.mv1.y == o.mv1.y && q.ref1 == o.ref1 && (q.dir == o.dir || q.dir == -1);
  if (!dup) out[n++] = tmp[1];
  return n;
}

// find_block_contexts (common_block.c:283-309)
TK_DEV void find_contexts(const DbCell* cells, int cs, int ypos, int xpos, int fh, int fw, int size, int enable,
                          SynCtx* s) {
  if (ypos >= kMinBlk && xpos >= kMinBlk && ypos + size < fh && xpos + size < fw && enable && size <= 128) {
    const int bi = (ypos / kMinPb) * cs + xpos / kMinPb;
    const DbCell& up = cells[bi - cs];
    const DbCell& le = cells[bi - 1];
    int split = (up.size < size) + (le.size < size);
    s->ctx_cbp = ((up.cbp & 1) != 0) + ((le.cbp & 1) != 0);
    int cbp2 = (up.cbp != 0) + (le.cbp != 0);
    s->ctx_index = 3 * split + cbp2;
  } else {
    s->ctx_cbp = -1;
    s->ctx_index = -1;
  }
}

// ---------------------------------------------------------------------------------
// SSD / cost
// ---------------------------------------------------------------------------------
// Sum of squared differences of two sample blocks, kept in registers: ssd_part() is this lane's share (8-bit samples: at most
// 256 samples x 255^2 per lane and 1.6e9 for the three planes of a 128x128 block - 32-bit arithmetic; 64-bit for 16-bit
// samples), ssd_total() the wave-wide sum - one DPP reduction per cost instead of an LDS accumulator round trip per plane.
// SP: address space of both sample blocks.
template <typename PIX> struct SsdT { typedef unsigned long long type; };
template <> struct SsdT<uint8_t> { typedef unsigned type; };
// 8-bit samples, four at a time: sum (a-b)^2 = sum a^2 + sum b^2 - 2 sum ab as three packed dot products (v_dot4_u32_u8) on the
// dwords as loaded; the partial sums wrap modulo 2^32 and the difference is exact (the true value fits, see above).
TK_DEV unsigned udot4_u8(unsigned a, unsigned b, unsigned c) {
#if TK_HOST
  for (int k = 0; k < 4; k++) c += ((a >> (8 * k)) & 0xffu) * ((b >> (8 * k)) & 0xffu);
  return c;
#else
  return __builtin_amdgcn_udot4(a, b, c, false);
#endif
}
TK_DEV unsigned udot2_u16(unsigned a, unsigned b, unsigned c) {
#if TK_HOST
  return c + (a & 0xffffu) * (b & 0xffffu) + (a >> 16) * (b >> 16);
#else
  typedef unsigned short __attribute__((ext_vector_type(2))) u16x2;
  u16x2 x, y;
  __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4);
  return __builtin_amdgcn_udot2(x, y, c, false);
#endif
}
// NW dwords per lane and step (4 * NW 8-bit or 2 * NW 16-bit samples); rows and pointers aligned to 4 * NW bytes.  16-bit samples (up
// to 12 bits): a lane's share is at most 256 samples x 4095^2 < 2^32, so the same modular arithmetic is exact.
template <int SP, typename PIX, int NW> TK_DEV unsigned ssd_rows(const Team t, const PIX* a_, int as, const PIX* b_, int bs, int w, int h) {
  const int spp = 4 * NW / (int)sizeof(PIX);   // samples per piece
  const int ppr = w / spp;
  const int lg = (ppr & (ppr - 1)) ? -1 : ilog2((unsigned)ppr);
  unsigned sq = 0, ab = 0;
  for (int k = t.rank; k < ppr * h; k += t.size) {
    int i, j;
    if (lg >= 0) { i = k >> lg; j = k & (ppr - 1); } else { i = k / ppr; j = k - i * ppr; }
    uint32_t x[NW], y[NW];
#if TK_HOST
    __builtin_memcpy(x, a_ + i * as + j * spp, 4 * NW);
    __builtin_memcpy(y, b_ + i * bs + j * spp, 4 * NW);
#else
    typedef uint32_t __attribute__((ext_vector_type(NW))) vec_t;
    const vec_t xv = *(typename SpT<SP, const vec_t>::ptr)(spc<SP>(a_) + i * as + j * spp);
    const vec_t yv = *(typename SpT<SP, const vec_t>::ptr)(spc<SP>(b_) + i * bs + j * spp);
    __builtin_memcpy(x, &xv, 4 * NW);
    __builtin_memcpy(y, &yv, 4 * NW);
#endif
#if !TK_HOST
#pragma unroll
#endif
    for (int q = 0; q < NW; q++) {
      if constexpr (sizeof(PIX) == 1) { sq = udot4_u8(x[q], x[q], udot4_u8(y[q], y[q], sq)); ab = udot4_u8(x[q], y[q], ab); }
      else { sq = udot2_u16(x[q], x[q], udot2_u16(y[q], y[q], sq)); ab = udot2_u16(x[q], y[q], ab); }
    }
  }
  return sq - 2u * ab;
}
template <int SP, typename PIX>
TK_DEV typename SsdT<PIX>::type ssd_part(const Team t, const PIX* a_, int as, const PIX* b_, int bs, int w, int h) {
  a_ = tk_uniform_ptr(a_); b_ = tk_uniform_ptr(b_); as = tk_uniform(as); bs = tk_uniform(bs); w = tk_uniform(w); h = tk_uniform(h);
#ifndef TK_NOVEC
  // (host simulation with teams smaller than a wavefront: a lane's share of a large block of 16-bit samples can exceed the 256 samples for
  // which the modular 32-bit sums of ssd_rows are exact - such blocks take the 64-bit sample loop below)
  if (!(TK_HOST && sizeof(PIX) == 2 && (w * h) / t.size > 256)) {
    const int S = (int)sizeof(PIX);
    const unsigned al = (unsigned)(uintptr_t)a_ | (unsigned)(uintptr_t)b_ | (unsigned)(as * S) | (unsigned)(bs * S) | (unsigned)(w * S);
    if (!(al & 15u)) return ssd_rows<SP, PIX, 4>(t, a_, as, b_, bs, w, h);
    if (!(al & 7u)) return ssd_rows<SP, PIX, 2>(t, a_, as, b_, bs, w, h);
    if (!(al & 3u)) return ssd_rows<SP, PIX, 1>(t, a_, as, b_, bs, w, h);
  }
#endif
  const auto a = spc<SP>(a_);
  const auto b = spc<SP>(b_);
  typename SsdT<PIX>::type local = 0;
  if ((w & (w - 1)) == 0) {  // every width except the frame-edge rectangles
    const Pow2 pw = mk_pow2(w);
    for (int k = t.rank; k < w * h; k += t.size) {
      int i, j;
      split2(pw, k, i, j);
      int d = (int)a[i * as + j] - (int)b[i * bs + j];
      local += (typename SsdT<PIX>::type)(d * d);
    }
  } else {
    for (int k = t.rank; k < w * h; k += t.size) {
      int i = k / w, j = k - i * w;
      int d = (int)a[i * as + j] - (int)b[i * bs + j];
      local += (typename SsdT<PIX>::type)(d * d);
    }
  }
  return local;
}
TK_DEV unsigned long long ssd_total(const Team t, unsigned v) { return (unsigned long long)(unsigned)team_sum(t, (int)v); }
TK_DEV unsigned long long ssd_total(const Team t, unsigned long long v) { return team_sum64(t, v); }

// cost_calc (encode_block.c:916-926) on the trial recon in ws->rec_* vs. the original frame.
template <typename PIX, int SP>
TK_DEVNI unsigned rd_cost(const Team t, JobR<PIX> J, WsP<PIX> ws, const Node& nd_, int nbits, double lambda,
                         long long ssd_y = -1) {
  TK_PROF_T0();
  const auto nd = ldsc(&nd_);
  const int size = TKU(nd->size), bw = TKU(nd->bw), bh = TKU(nd->bh);
  const int sc = size >> 1;
  typename SsdT<PIX>::type part = 0;
  if (ssd_y < 0) part += ssd_part<SP>(t, ws->org_y, ws->org_sy, ws->rec_y, size, bw, bh);
  part += ssd_part<SP>(t, ws->org_u, ws->org_sc, ws->rec_u, sc, bw >> 1, bh >> 1);
  part += ssd_part<SP>(t, ws->org_v, ws->org_sc, ws->rec_v, sc, bw >> 1, bh >> 1);
  const unsigned long long ssd = ssd_total(t, part) + (ssd_y >= 0 ? (unsigned long long)ssd_y : 0ull);
  unsigned long long cost = (ssd >> (J.cfg.bitdepth * 2 - 16)) + (unsigned long long)(long long)mul_add_nofma(lambda, (double)nbits, 0.5);
  if (cost > (1ull << 30)) cost = 1ull << 30;
  TK_PROF_ADD(ws, PF_COST);
  return (unsigned)cost;
}

// ---------------------------------------------------------------------------------
// Chroma-from-luma (common_block.c:347-428).  y: luma prediction (stride n), u/v: chroma
// prediction (stride cstride>>1), ry: reconstructed luma (stride `stride`), n = luma size.
// ---------------------------------------------------------------------------------
template <typename PIX, int SP>
TK_DEVNI void improve_uv(const Team t, WsP<PIX> ws, const PIX* y_, PIX* u_, PIX* v_, const PIX* ry_, int n, int cstride,
                       int stride, int bitdepth) {
  const auto y = spc<SP>(y_); const auto u = spc<SP>(u_); const auto v = spc<SP>(v_); const auto ry = spc<SP>(ry_);
  (void)ws;
  const int nc = n >> 1, lognc = ilog2(nc), cs = cstride >> 1;
  long long tot8[8];
  typedef typename SsdT<PIX>::type sum_t;   // 32-bit sums for 8-bit samples (at most 4096 x 255^2 per sum), 64-bit otherwise
  long long sq;
  {
    sum_t local = 0;
    for (int k = t.rank; k < n * n; k += t.size) {
      int i, j;
      split2(mk_pow2(n), k, i, j);
      int d = (int)ry[i * stride + j] - (int)y[i * n + j];
      local += (sum_t)(d * d);
    }
    sq = (long long)ssd_total(t, local);
  }
  if ((sq >> (2 * ilog2(n))) <= (64ll << (2 * (bitdepth - 8)))) { t.sync(); return; }
  {
    sum_t ls[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = t.rank; k < nc * nc; k += t.size) {
      int i, j;
      split2(mk_pow2(nc), k, i, j);
      int us = u[i * cs + j], vs = v[i * cs + j];
      int ys = (y[(i * 2) * n + j * 2] + y[(i * 2) * n + j * 2 + 1] + y[(i * 2 + 1) * n + j * 2] + y[(i * 2 + 1) * n + j * 2 + 1] + 2) >> 2;
      ls[0] += ys; ls[1] += us; ls[2] += vs;
      ls[3] += (unsigned)(ys * ys); ls[4] += (unsigned)(ys * us); ls[5] += (unsigned)(ys * vs);
      ls[6] += (unsigned)(us * us); ls[7] += (unsigned)(vs * vs);
    }
    for (int q = 0; q < 8; q++) ls[q] = (sum_t)ssd_total(t, ls[q]);
    tot8[0] = (long long)ls[0]; tot8[1] = (long long)ls[1]; tot8[2] = (long long)ls[2]; tot8[3] = (long long)ls[3];
    tot8[4] = (long long)ls[4]; tot8[5] = (long long)ls[5]; tot8[6] = (long long)ls[6]; tot8[7] = (long long)ls[7];
  }
  const long long ysum = tot8[0], usum = tot8[1], vsum = tot8[2], yysum = tot8[3], yusum = tot8[4],
                  yvsum = tot8[5], uusum = tot8[6], vvsum = tot8[7];
  const long long ssyy = yysum - ((ysum * ysum) >> (lognc * 2));
  const long long ssuu = uusum - ((usum * usum) >> (lognc * 2));
  const long long ssvv = vvsum - ((vsum * vsum) >> (lognc * 2));
  const long long ssyu = yusum - ((ysum * usum) >> (lognc * 2));
  const long long ssyv = yvsum - ((ysum * vsum) >> (lognc * 2));
  if (!ssyy) return;
  for (int pl = 0; pl < 2; pl++) {
    const long long ssyc = pl ? ssyv : ssyu, sscc = pl ? ssvv : ssuu, csum = pl ? vsum : usum;
    const auto c = pl ? v : u;
    if (ssyc * ssyc * 2 > ssyy * sscc) {
      long long a64 = (ssyc << 16) / ssyy;
      long long b64 = ((csum << 16) - a64 * ysum) >> (lognc * 2);
      const long long alim = 1ll << (31 - bitdepth);
      int a = (int)(a64 < -alim ? -alim : (a64 > alim ? alim : a64));
      long long bb = b64 + (1 << 15);
      int b = (int)(bb < -(1ll << 31) ? -(1ll << 31) : (bb > ((1ll << 31) - 1) ? ((1ll << 31) - 1) : bb));
      for (int k = t.rank; k < nc * nc; k += t.size) {
        int i, j;
        split2(mk_pow2(nc), k, i, j);
        int s = 2;
        for (int q = 0; q < 4; q++) {
          int r = ry[(i * 2 + (q >> 1)) * stride + j * 2 + (q & 1)];
          int m = (int)((unsigned)a * (unsigned)r + (unsigned)b);  // wraps like the reference's int arithmetic
          s += sat_pix(m >> 16, bitdepth);
        }
        c[i * cs + j] = (PIX)(s >> 2);
      }
    }
  }
  t.sync();
}

// ---------------------------------------------------------------------------------
// encode_block (encode_block.c:1340-1514): prediction + residual coding of one CB into the trial
// buffers ws->rec_* / ws->coef_*; returns the number of bits of write_block.  `bs` counts or emits.
// ---------------------------------------------------------------------------------
template <typename PIX, int SP>
TK_DEV void predict_inter(const Team t, JobR<PIX> J, WsP<PIX> ws, const Node& nd_, const BlkParam& p,
                          int split) {
  const auto ndl = ldsc(&nd_);
  struct { int ypos, xpos, size, bw, bh; } nd = {TKU(ndl->ypos), TKU(ndl->xpos), TKU(ndl->size), TKU(ndl->bw), TKU(ndl->bh)};
  TK_PROF_T0();
  const auto& c = J.cfg;
  const int bi = (p.mode == M_BIPRED) || ((p.mode == M_SKIP || p.mode == M_MERGE) && p.dir == 2);
  if (bi) {
    pred_inter_yuv<SP>(t, lds_ld(&J.ref[p.ref0]), ws->p0_y, ws->p0_u, ws->p0_v, nd.ypos, nd.xpos, nd.size, nd.bw, nd.bh, p.mv0,
                   J.sign[p.ref0], c.width, c.height, c.enable_bipred, split, c.bitdepth);
    pred_inter_yuv<SP>(t, lds_ld(&J.ref[p.ref1]), ws->p1_y, ws->p1_u, ws->p1_v, nd.ypos, nd.xpos, nd.size, nd.bw, nd.bh, p.mv1,
                   J.sign[p.ref1], c.width, c.height, c.enable_bipred, split, c.bitdepth);
    t.sync();
    average_yuv<SP>(t, ws->pred_y, ws->pred_u, ws->pred_v, ws->p0_y, ws->p0_u, ws->p0_v, ws->p1_y, ws->p1_u, ws->p1_v,
                nd.size, nd.bw, nd.bh);
  } else {
    pred_inter_yuv<SP>(t, lds_ld(&J.ref[p.ref0]), ws->pred_y, ws->pred_u, ws->pred_v, nd.ypos, nd.xpos, nd.size, nd.bw, nd.bh,
                   p.mv0, J.sign[p.ref0], c.width, c.height, c.enable_bipred, split, c.bitdepth);
  }
  t.sync();
  TK_PROF_ADD(ws, PF_PRED_INTER);
}

// Exact partial-cost pruning of RDO trials.  A trial only matters if its cost is below a threshold the caller
// knows (the best cost so far; for the intra search also the best intra cost so far - strict '<' everywhere in
// mode_decision_rdo).  cost = SSD_Y + SSD_U + SSD_V + (unsigned)(lambda * bits + 0.5) is monotone in every
// term, so once the luma planes are coded, SSD_Y + (unsigned)(lambda * luma coefficient bits + 0.5) is a lower
// bound of the final cost: if it already reaches the threshold the chroma transform units, CfL, the bit count
// and the cost evaluation are skipped and the trial is reported as "not better" - results are unchanged.
// Parallel decision (mode_decision_par): the trials of one block run on several wavefronts in no particular order,
// so the threshold is the shared minimum over all FINISHED trials of the key (cost << 32 | evaluation order) - the
// winner is the trial with the smallest key, which is exactly the reference's "first strictly smaller cost in
// evaluation order".  A trial whose lower-bound key (lb << 32 | its order) already exceeds that minimum cannot have
// the smallest key, whatever the timing: pruning stays exact and only the amount of skipped work varies.
struct PruneCtx {
  unsigned thr;       // prune when the lower bound is >= thr (0xffffffff: never)
  const unsigned long long* bestkey;  // parallel decision: shared minimum key (nullptr: use thr)
  unsigned order;                     // evaluation order of this trial
  double lambda;
  long long ssd_y;    // out: luma SSD of the trial (reused by rd_cost), -1 if not computed
  int ybits[4];       // out: luma coefficient bits per TU
  int have_ybits;
  int pruned;         // out
  long long ssd_part; // tb-split luma: SSD / bits of the quadrants coded so far
  int bits_part;
  int head_bits;      // bits of the trial that do not depend on its residual (tk_bits.h:bs_block_head_t): part of every bound
};

// tb-split luma: call after quadrant `tu` (0..3, size s2 at (i,j) of the block) has been coded.  The first three
// quadrants give an early lower bound; after the fourth the accumulated values are the block's luma SSD / bits.
TK_DEV int prune_active(const PruneCtx* pc) { return pc && (pc->bestkey || pc->thr != 0xffffffffu); }
TK_DEV int prune_hit(const PruneCtx* pc, unsigned long long lb) {
  if (pc->bestkey) return ((lb << 32) | (unsigned long long)pc->order) > wg_load64(pc->bestkey);
  return lb >= (unsigned long long)pc->thr;
}

template <typename PIX, int SP>
TK_DEV int prune_after_quadrant(const Team t, JobR<PIX> J, WsP<PIX> ws, int nd_size, int intra, int tu, int i, int j,
                                int s2, int bit, const int16_t* coef, PruneCtx* pc) {
  if (!prune_active(pc)) return 0;
  t.sync();
  pc->ssd_part += (long long)ssd_total(t, ssd_part<SP>(t, ws->org_y + i * ws->org_sy + j, ws->org_sy, ws->rec_y + i * nd_size + j, nd_size, s2, s2));
  pc->ybits[tu] = bit ? coeff_bits_team<SP_LDS>(t, coef, s2, intra << 1) : 0;  // luma coefficients: always SmallWs (LDS)
  pc->bits_part += pc->ybits[tu];
  if (tu == 3) { pc->ssd_y = pc->ssd_part; pc->have_ybits = 1; }
  unsigned long long lb = ((unsigned long long)pc->ssd_part >> (J.cfg.bitdepth * 2 - 16)) + (unsigned long long)(long long)mul_add_nofma(pc->lambda, (double)(pc->bits_part + pc->head_bits), 0.5);
  if (lb > (1ull << 30)) lb = 1ull << 30;
#if TK_HOST
  { extern long long g_prune_stat[8]; g_prune_stat[4] += 1; if (prune_hit(pc, lb)) g_prune_stat[5 + (tu == 3)] += 1; }
#endif
  if (team_bcast0(t, prune_hit(pc, lb))) { pc->pruned = 1; return 1; }  // one lane's reading decides for the wave
  return 0;
}

template <typename PIX, int SP>
TK_DEV int prune_after_luma(const Team t, JobR<PIX> J, WsP<PIX> ws, int size, int bw, int bh, const BlkParam& p, int cbp_y,
                            int tb_split, PruneCtx* pc) {
  if (!prune_active(pc)) return 0;
  if (pc->pruned) return 1;
  if (pc->have_ybits) return 0;  // tb-split luma: bound already evaluated quadrant by quadrant
  t.sync();
  const unsigned long long ssd = ssd_total(t, ssd_part<SP>(t, ws->org_y, ws->org_sy, ws->rec_y, size, bw, bh));
  pc->ssd_y = (long long)ssd;
  const int coeff_type = (p.mode == M_INTRA) << 1;
  int bits = 0;
  if (!tb_split) {
    pc->ybits[0] = cbp_y ? coeff_bits_team<SP_LDS>(t, ws->coef_y, size, coeff_type) : 0;
    bits = pc->ybits[0];
  } else {
    const int qy = size / 2 < kMaxQuant ? size / 2 : kMaxQuant;
    for (int tu = 0; tu < 4; tu++) {
      pc->ybits[tu] = ((cbp_y >> (3 - tu)) & 1) ? coeff_bits_team<SP_LDS>(t, ws->coef_y + tu * qy * qy, size / 2, coeff_type) : 0;
      bits += pc->ybits[tu];
    }
  }
  pc->have_ybits = 1;
  unsigned long long lb = (ssd >> (J.cfg.bitdepth * 2 - 16)) + (unsigned long long)(long long)mul_add_nofma(pc->lambda, (double)(bits + pc->head_bits), 0.5);
  if (lb > (1ull << 30)) lb = 1ull << 30;
#if TK_HOST
  { extern long long g_prune_stat[8]; g_prune_stat[p.mode == M_INTRA ? 0 : 2] += 1; if (prune_hit(pc, lb)) g_prune_stat[p.mode == M_INTRA ? 1 : 3] += 1; }
#endif
  if (team_bcast0(t, prune_hit(pc, lb))) { pc->pruned = 1; return 1; }  // one lane's reading decides for the wave
  return 0;
}

// residual coding of one plane of an inter block (encode_and_reconstruct_block_inter :1275-1338)
// SP: address space of org / pred / rec, SC: of coef.  nd_size > 0: luma plane of a block of that size with pruning context pc.
template <typename PIX, int SP, int SC>
TK_DEV int code_inter_plane(const Team t, JobR<PIX> J, WsP<PIX> ws, const PIX* org, int ostride,
                            const PIX* pred, PIX* rec, int size, int qp, int coeff_type, int tb_split, int16_t* coef,
                            int nd_size = 0, PruneCtx* pc = nullptr) {
  const int bd = J.cfg.bitdepth;
  if (!tb_split) {
    int fast = (size == 64 && J.cfg.encoder_speed > 0) || J.cfg.encoder_speed > 1;
    return code_tu_sp<PIX, SP, SC>(t, ws->xfp, org, ostride, pred, size, rec, size, size, qp, coeff_type, fast, coef, bd);
  }
  const int s2 = size / 2;
  int cbp = 0, index = 0;
  for (int i = 0; i < size; i += s2)
    for (int j = 0; j < size; j += s2) {
      int fast = size == 64 || J.cfg.encoder_speed > 1;
      int bit = code_tu_sp<PIX, SP, SC>(t, ws->xfp, org + i * ostride + j, ostride, pred + i * size + j, size, rec + i * size + j, size,
                        s2, qp, coeff_type, fast, coef + index, bd);
      cbp = (cbp << 1) + bit;
      if (nd_size && prune_after_quadrant<PIX, SP>(t, J, ws, nd_size, 0, (i ? 2 : 0) + (j ? 1 : 0), i, j, s2, bit, coef + index, pc)) return cbp;
      index += tmin(s2, 16) * tmin(s2, 16);
    }
  return cbp;
}

// reuse_pred: the inter prediction of this (mode, refs, MVs) is already in ws->pred_* (previous trial
// of the same candidate with another tb_param) - exact, the prediction does not depend on tb_param.
// reuse_pred == 2: the caller vouches that ws->pred_* holds the prediction untouched (no CfL pass has refined its chroma).
// SP: address space of the coding block's sample buffers and original samples (SP_LDS for blocks up to kLdsBlk).
template <typename PIX, int SP>
TK_DEVNI int encode_block(const Team t, JobR<PIX> J, WsP<PIX> ws, Node& nd_, BlkParam& p, BitSink& bs,
                          int reuse_pred = 0, PruneCtx* pc = nullptr) {
  const auto& c = J.cfg;
  const auto ndl = ldsc(&nd_);
  struct { int size, ypos, xpos, bw, bh; } nd = {TKU(ndl->size), TKU(ndl->ypos), TKU(ndl->xpos), TKU(ndl->bw), TKU(ndl->bh)};
  const int size = nd.size, sizeC = size >> 1;
  const int yc = nd.ypos >> 1, xc = nd.xpos >> 1;
  const int qpY = TKU(J.qp), qpC = TK_TAB.chroma_qp[qpY];
  const int tb_split = TKU(p.tb_param) > 0 ? TKU(p.tb_param) : 0;
  const int zero_block = TKU(p.tb_param) == -1;
  const int ftI = (TKU(J.frame_type) == F_I) << 1;
  const int bd = TKU(c.bitdepth);
  p.tb_split = (int8_t)tb_split;
  // chroma coefficients: SmallWs (LDS) except the 4 x 16x16 units of tb-split 64 / 128 blocks (global scratch)
  const int bigc = SP == SP_GLOBAL && tb_split && sizeC >= 32;
  ws->coef_u = bigc ? ws->coef_u_big : ws->coef_u_small;
  ws->coef_v = bigc ? ws->coef_v_big : ws->coef_v_small;
  const PIX* oy = ws->org_y;
  const PIX* ou = ws->org_u;
  const PIX* ov = ws->org_v;
  const int osy = TKU(ws->org_sy), osc = TKU(ws->org_sc);
  int cbp_y = 0, cbp_u = 0, cbp_v = 0;

  if (TKU(p.mode) == M_INTRA) {
    const int ur = upright_avail(nd.ypos, nd.xpos, size, size, c.width, kMaxSb);
    const int dl = downleft_avail(nd.ypos, nd.xpos, size, size, c.height, kMaxSb);
    const PIX* fy = J.rec.y + nd.ypos * J.rec.sy + nd.xpos;
    const PIX* fu = J.rec.u + yc * J.rec.sc + xc;
    const PIX* fv = J.rec.v + yc * J.rec.sc + xc;
    // luma (encode_and_reconstruct_block_intra :1100-1168)
    if (tb_split) {
      const int s2 = size / 2;
      int index = 0;
      for (int i = 0; i < size; i += s2)
        for (int j = 0; j < size; j += s2) {
          make_edges<SP>(t, ws->edgep, fy, J.rec.sy, ws->rec_y + i * size + j, size, i, j, nd.ypos, nd.xpos, s2, ur, dl, 1, bd);
          pred_intra<SP>(t, ws->edgep, nd.ypos + i, nd.xpos + j, s2, ws->pred_y + i * size + j, size, p.intra_mode, bd);
          t.sync();
          int bit = code_tu_sp<PIX, SP, SP_LDS>(t, ws->xfp, oy + i * osy + j, osy, ws->pred_y + i * size + j, size,
                            ws->rec_y + i * size + j, size, s2, qpY, ftI | 0, c.encoder_speed > 1, ws->coef_y + index, bd);
          cbp_y = (cbp_y << 1) + bit;
          if (prune_after_quadrant<PIX, SP>(t, J, ws, size, 1, (i ? 2 : 0) + (j ? 1 : 0), i, j, s2, bit, ws->coef_y + index, pc)) return 0;
          index += tmin(s2, 16) * tmin(s2, 16);
        }
    } else {
      make_edges<SP>(t, ws->edgep, fy, J.rec.sy, (const PIX*)nullptr, 0, 0, 0, nd.ypos, nd.xpos, size, ur, dl, 0, bd);
      pred_intra<SP>(t, ws->edgep, nd.ypos, nd.xpos, size, ws->pred_y, size, p.intra_mode, bd);
      t.sync();
      cbp_y = code_tu_sp<PIX, SP, SP_LDS>(t, ws->xfp, oy, osy, ws->pred_y, size, ws->rec_y, size, size, qpY, ftI | 0,
                      c.encoder_speed > 1, ws->coef_y, bd);
    }
    if (prune_after_luma<PIX, SP>(t, J, ws, size, nd.bw, nd.bh, p, cbp_y, tb_split, pc)) return 0;
    // chroma (encode_and_reconstruct_block_intra_uv :1170-1273)
    const int csplit = tb_split && sizeC > 4;
    if (csplit) {
      const int s2 = sizeC / 2;
      int index = 0;
      for (int i = 0; i < sizeC; i += s2)
        for (int j = 0; j < sizeC; j += s2) {
          make_edges<SP>(t, ws->edgep, fu, J.rec.sc, ws->rec_u + i * sizeC + j, sizeC, i, j, yc, xc, s2, ur, dl, 1, bd);
          pred_intra<SP>(t, ws->edgep, yc + i, xc + j, s2, ws->pred_u + i * sizeC + j, sizeC, p.intra_mode, bd);
          t.sync();
          make_edges<SP>(t, ws->edgep, fv, J.rec.sc, ws->rec_v + i * sizeC + j, sizeC, i, j, yc, xc, s2, ur, dl, 1, bd);
          pred_intra<SP>(t, ws->edgep, yc + i, xc + j, s2, ws->pred_v + i * sizeC + j, sizeC, p.intra_mode, bd);
          t.sync();
          if (c.cfl_intra)  // sic: luma pointers offset in CHROMA units (encode_block.c:1199)
            improve_uv<PIX, SP>(t, ws, ws->pred_y + i * sizeC + j, ws->pred_u + i * sizeC + j, ws->pred_v + i * sizeC + j,
                       ws->rec_y + (i << 1) * size + (j << 1), s2 << 1, sizeC << 1, size, bd);
          int bu, bv;
          if (bigc) {
            bu = code_tu_sp<PIX, SP, SP_GLOBAL>(t, ws->xfp, ou + i * osc + j, osc, ws->pred_u + i * sizeC + j, sizeC,
                             ws->rec_u + i * sizeC + j, sizeC, s2, qpC, ftI | 1, c.encoder_speed > 1, ws->coef_u + index, bd);
            bv = code_tu_sp<PIX, SP, SP_GLOBAL>(t, ws->xfp, ov + i * osc + j, osc, ws->pred_v + i * sizeC + j, sizeC,
                             ws->rec_v + i * sizeC + j, sizeC, s2, qpC, ftI | 1, c.encoder_speed > 1, ws->coef_v + index, bd);
          } else {
            bu = code_tu_sp<PIX, SP, SP_LDS>(t, ws->xfp, ou + i * osc + j, osc, ws->pred_u + i * sizeC + j, sizeC,
                             ws->rec_u + i * sizeC + j, sizeC, s2, qpC, ftI | 1, c.encoder_speed > 1, ws->coef_u + index, bd);
            bv = code_tu_sp<PIX, SP, SP_LDS>(t, ws->xfp, ov + i * osc + j, osc, ws->pred_v + i * sizeC + j, sizeC,
                             ws->rec_v + i * sizeC + j, sizeC, s2, qpC, ftI | 1, c.encoder_speed > 1, ws->coef_v + index, bd);
          }
          cbp_u = (cbp_u << 1) + bu;
          cbp_v = (cbp_v << 1) + bv;
          index += tmin(s2, 16) * tmin(s2, 16);
        }
    } else {
      make_edges<SP>(t, ws->edgep, fu, J.rec.sc, (const PIX*)nullptr, 0, 0, 0, yc, xc, sizeC, ur, dl, 0, bd);
      pred_intra<SP>(t, ws->edgep, yc, xc, sizeC, ws->pred_u, sizeC, p.intra_mode, bd);
      t.sync();
      make_edges<SP>(t, ws->edgep, fv, J.rec.sc, (const PIX*)nullptr, 0, 0, 0, yc, xc, sizeC, ur, dl, 0, bd);
      pred_intra<SP>(t, ws->edgep, yc, xc, sizeC, ws->pred_v, sizeC, p.intra_mode, bd);
      t.sync();
      if (c.cfl_intra) improve_uv<PIX, SP>(t, ws, ws->pred_y, ws->pred_u, ws->pred_v, ws->rec_y, size, size, size, bd);
      cbp_u = code_tu_sp<PIX, SP, SP_LDS>(t, ws->xfp, ou, osc, ws->pred_u, sizeC, ws->rec_u, sizeC, sizeC, qpC, ftI | 1,
                      c.encoder_speed > 1, ws->coef_u, bd);
      cbp_v = code_tu_sp<PIX, SP, SP_LDS>(t, ws->xfp, ov, osc, ws->pred_v, sizeC, ws->rec_v, sizeC, sizeC, qpC, ftI | 1,
                      c.encoder_speed > 1, ws->coef_v, bd);
    }
  } else {
    const int split = (TKU(p.mode) == M_INTER || TKU(p.mode) == M_BIPRED) ? c.enable_pb_split : 0;
    if (!(reuse_pred == 2 || (reuse_pred && !c.cfl_inter))) predict_inter<PIX, SP>(t, J, ws, nd_, p, split);
    if (TKU(p.mode) == M_SKIP || zero_block) {
      copy_block<SP, SP>(t, ws->rec_y, size, ws->pred_y, size, nd.bw, nd.bh);
      copy_block<SP, SP>(t, ws->rec_u, sizeC, ws->pred_u, sizeC, nd.bw >> 1, nd.bh >> 1);
      copy_block<SP, SP>(t, ws->rec_v, sizeC, ws->pred_v, sizeC, nd.bw >> 1, nd.bh >> 1);
      t.sync();
    } else {
      cbp_y = code_inter_plane<PIX, SP, SP_LDS>(t, J, ws, oy, osy, ws->pred_y, ws->rec_y, size, qpY, ftI | 0, tb_split, ws->coef_y, size, pc);
      if (prune_after_luma<PIX, SP>(t, J, ws, size, nd.bw, nd.bh, p, cbp_y, tb_split, pc)) return 0;
      if (c.cfl_inter) improve_uv<PIX, SP>(t, ws, ws->pred_y, ws->pred_u, ws->pred_v, ws->rec_y, size, size, size, bd);
      const int csplit = tb_split && sizeC > 4;
      if (bigc) {
        cbp_u = code_inter_plane<PIX, SP, SP_GLOBAL>(t, J, ws, ou, osc, ws->pred_u, ws->rec_u, sizeC, qpC, ftI | 1, csplit, ws->coef_u);
        cbp_v = code_inter_plane<PIX, SP, SP_GLOBAL>(t, J, ws, ov, osc, ws->pred_v, ws->rec_v, sizeC, qpC, ftI | 1, csplit, ws->coef_v);
      } else {
        cbp_u = code_inter_plane<PIX, SP, SP_LDS>(t, J, ws, ou, osc, ws->pred_u, ws->rec_u, sizeC, qpC, ftI | 1, csplit, ws->coef_u);
        cbp_v = code_inter_plane<PIX, SP, SP_LDS>(t, J, ws, ov, osc, ws->pred_v, ws->rec_v, sizeC, qpC, ftI | 1, csplit, ws->coef_v);
      }
    }
  }
  p.cbp_y = (uint8_t)cbp_y;
  p.cbp_u = (uint8_t)cbp_u;
  p.cbp_v = (uint8_t)cbp_v;
  TK_PROF_T0();
  const SynCtx syn = lds_ld(&nd_.syn);
  const int* yb = (pc && pc->have_ybits) ? pc->ybits : nullptr;
  int nb_ = bigc ? bs_block_t<false, SP_GLOBAL>(bs, syn, p, ws->coef_y, ws->coef_u, ws->coef_v, &t, yb)   // bs.emit == 0 always here
                 : bs_block_t<false, SP_LDS>(bs, syn, p, ws->coef_y, ws->coef_u, ws->coef_v, &t, yb);
  TK_PROF_ADD(ws, PF_BITS);
  return nb_;
}

// One RDO trial: count bits, evaluate cost, keep `best` (copy_best_parameters, :1615-1677).
template <typename PIX, int SP>
TK_DEV unsigned rdo_trial(const Team t, JobR<PIX> J, WsP<PIX> ws, Node& nd, BlkParam& p, double lambda,
                          int reuse_pred = 0, unsigned prune_thr = 0xffffffffu, const unsigned long long* bestkey = nullptr,
                          unsigned order = 0, int* nbits_out = nullptr, int* untouched = nullptr) {
  BitSink cnt;
  cnt.buf = nullptr; cnt.pos = 0; cnt.cap = 0; cnt.emit = 0; cnt.ovf = 0;
  PruneCtx pc;
  pc.thr = prune_thr; pc.bestkey = bestkey; pc.order = order; pc.lambda = lambda; pc.ssd_y = -1; pc.have_ybits = 0; pc.pruned = 0; pc.ssd_part = 0; pc.bits_part = 0;
  pc.head_bits = 0;
  if (bestkey || prune_thr != 0xffffffffu) {
    // The bits that do not depend on the residual (super-mode, partition, vector differences, intra mode, candidate index) are
    // known before anything is predicted or transformed, and the cost is monotone in every term: a trial whose bound with
    // SSD = 0 and no other bits already reaches the threshold / exceeds the shared minimum is dropped before it starts
    // (`untouched`: the prediction buffers still hold what they held), and the later bounds (luma coded) start from these
    // bits.  (A later trial of the same candidate has the same header bits against a threshold that has not grown: it is
    // dropped the same way and never asks for the prediction an earlier one did not build.)
    BitSink hb = cnt;
    bs_block_head_t<false>(hb, uniform_syn(lds_ld(&nd.syn)), uniform_blk(p));
    pc.head_bits = hb.pos;
    unsigned long long lb0 = (unsigned long long)(long long)mul_add_nofma(lambda, (double)hb.pos, 0.5);
    if (lb0 > (1ull << 30)) lb0 = 1ull << 30;
#if TK_HOST
    { extern long long g_prune_stat[8]; g_prune_stat[7] += prune_hit(&pc, lb0); }
#endif
    if (team_bcast0(t, prune_hit(&pc, lb0))) {
      if (untouched) *untouched = 1;
      return kCostInit;
    }
  }
  int nbits = encode_block<PIX, SP>(t, J, ws, nd, p, cnt, reuse_pred, &pc);
  if (nbits_out) *nbits_out = nbits;
  if (pc.pruned) return kCostInit;  // lower bound >= threshold: cannot be selected
  return rd_cost<PIX, SP>(t, J, ws, nd, nbits, lambda, pc.ssd_y);
}

TK_DEV BlkParam normalize_best(const Node& nd, const BlkParam& p) {
  BlkParam b = p;
  if (p.mode == M_SKIP || p.mode == M_MERGE) {
    const InterPred c = lds_ld((p.mode == M_SKIP) ? &nd.skip[p.skip_idx] : &nd.merge[p.skip_idx]);
    b.ref0 = c.ref0; b.ref1 = c.ref1; b.dir = c.dir;
    for (int i = 0; i < 4; i++) { b.mv0[i] = c.mv0; b.mv1[i] = c.mv1; }
  } else if (p.mode == M_INTRA) {
    b.ref0 = b.ref1 = 0; b.dir = -1;
    for (int i = 0; i < 4; i++) { b.mv0[i] = mk_mv(0, 0); b.mv1[i] = mk_mv(0, 0); }
  } else if (p.mode == M_INTER) b.dir = 0;
  else b.dir = 2;
  return b;
}
TK_DEV void keep_best(Node& nd, const BlkParam& p) { lds_st(&nd.best, normalize_best(nd, p)); }

TK_DEV void set_cand(BlkParam& p, const InterPred& c, int idx, int mode) {
  p.mode = (int8_t)mode;
  p.skip_idx = (int8_t)idx;
  p.ref0 = c.ref0; p.ref1 = c.ref1; p.dir = c.dir;
  for (int i = 0; i < 4; i++) { p.mv0[i] = c.mv0; p.mv1[i] = c.mv1; }
}

// search_inter_prediction_params (encode_block.c:1033-1098)
// SP: address space of `org`
template <typename PIX, int SP>
TK_DEV unsigned search_inter(const Team t, JobR<PIX> J, WsP<PIX> ws, int ypos, int xpos, int size,
                             const PIX* org, int ostride, int ref_idx, mv_t mvc, mv_t mvp, mv_t* mv_arr, int part,
                             int sign) {
  const Plane3<PIX> ref = lds_ld(&J.ref[ref_idx]);
  const PIX* ref_y = ref.y + ypos * ref.sy + xpos;
  MeArgs a;
  a.cb_size = size; a.rstride = ref.sy; a.sign = sign; a.fwidth = J.cfg.width; a.fheight = J.cfg.height;
  a.xpos = xpos; a.ypos = ypos; a.enable_bipred = J.cfg.enable_bipred; a.bitdepth = J.cfg.bitdepth;
  a.lam = J.sqrt_lambda; a.ostride = ostride; a.speed = J.cfg.encoder_speed;
  unsigned sad = 0;
  mv_t mv, mvp2 = mvp;
  if (part == P_NONE) {
    a.width = size; a.height = size; a.pu_x = xpos; a.pu_y = ypos;
    sad += motion_estimate<PIX, SP>(t, ws->mep, org, ref_y, a, mvc, mvp2, ref_idx, &mv);
    mv_arr[0] = mv_arr[1] = mv_arr[2] = mv_arr[3] = mv;
  } else if (part == P_HOR) {
    a.width = size; a.height = size / 2;
    for (int index = 0; index < 4; index += 2) {
      int py = index >> 1;
      a.pu_x = xpos; a.pu_y = ypos + py * (size / 2);
      sad += motion_estimate<PIX, SP>(t, ws->mep, org + py * (size / 2) * ostride, ref_y + py * (size / 2) * ref.sy, a, mvc, mvp2, ref_idx, &mv);
      mv_arr[index] = mv; mv_arr[index + 1] = mv;
      mvp2 = mv_arr[0];
    }
  } else if (part == P_VER) {
    a.width = size / 2; a.height = size;
    for (int index = 0; index < 2; index++) {
      a.pu_x = xpos + index * (size / 2); a.pu_y = ypos;
      sad += motion_estimate<PIX, SP>(t, ws->mep, org + index * (size / 2), ref_y + index * (size / 2), a, mvc, mvp2, ref_idx, &mv);
      mv_arr[index] = mv; mv_arr[index + 2] = mv;
      mvp2 = mv_arr[0];
    }
  } else {
    a.width = size / 2; a.height = size / 2;
    for (int index = 0; index < 4; index++) {
      int px = index & 1, py = index >> 1;
      a.pu_x = xpos + px * (size / 2); a.pu_y = ypos + py * (size / 2);
      sad += motion_estimate<PIX, SP>(t, ws->mep, org + py * (size / 2) * ostride + px * (size / 2),
                             ref_y + py * (size / 2) * ref.sy + px * (size / 2), a, mvc, mvp2, ref_idx, &mv);
      mv_arr[index] = mv;
      mvp2 = mv_arr[0];
    }
  }
  return sad;
}

template <typename PIX> TK_DEV void add_cands4(const Team t, WsP<PIX> ws, int ref_idx, const mv_t* mv4) {
  if (t.rank == 0)
    for (int i = 0; i < 4; i++) add_mvcand(ws->mep, ref_idx, mv4[i]);
  t.sync();
}

// 2 * org - pred, saturated (the "original" of a bi-prediction search step, encode_block.c:1786-1791), for a size x size block:
// four samples per lane and step (sample blocks and original rows are aligned to four samples), sample by sample otherwise.
template <typename PIX, int SP>
TK_DEV void build_org8(const Team t, PIX* o8_, const PIX* oy_, int osy, const PIX* py_, int size, int bitdepth) {
#ifndef TK_NOVEC
  const int S = (int)sizeof(PIX);
  const unsigned al = (unsigned)(uintptr_t)o8_ | (unsigned)(uintptr_t)oy_ | (unsigned)(uintptr_t)py_ | (unsigned)(osy * S);
  if (tk_uniform(!(al & (unsigned)(4 * S - 1)))) {
    const int ppr = size >> 2, lg = ilog2((unsigned)ppr);
    for (int k = t.rank; k < ppr * size; k += t.size) {
      const int i = k >> lg, j = (k & (ppr - 1)) << 2;
      int o[4], p[4];
      load_samples<SP, PIX, 4>(oy_ + i * osy + j, o);
      load_samples<SP, PIX, 4>(py_ + i * size + j, p);
      for (int q = 0; q < 4; q++) o[q] = sat_pix(2 * o[q] - p[q], bitdepth);
      store_samples<SP, PIX, 4>(o8_ + i * size + j, o);
    }
    return;
  }
#endif
  const auto o8 = spc<SP>(o8_);
  const auto oys = spc<SP>(oy_);
  const auto pys = spc<SP>(py_);
  for (int k = t.rank; k < size * size; k += t.size) {
    int i, j;
    split2(mk_pow2(size), k, i, j);
    o8[k] = (PIX)sat_pix(2 * (int)oys[i * osy + j] - (int)pys[k], bitdepth);
  }
}

// search_bipred_prediction_params, me_mode 0 (encode_block.c:1739-1832) - P and B frames.
template <typename PIX, int SP>
TK_DEVNI void search_bipred(const Team t, JobR<PIX> J, WsP<PIX> ws, const Node& nd_, int part,
                          const mv_t* mv_center, mv_t mvp, int* ref_idx0, int* ref_idx1, mv_t* mv_arr0, mv_t* mv_arr1) {
  const auto& c = J.cfg;
  const auto ndl = ldsc(&nd_);
  struct { int size, ypos, xpos, bw, bh; } nd = {TKU(ndl->size), TKU(ndl->ypos), TKU(ndl->xpos), TKU(ndl->bw), TKU(ndl->bh)};
  const auto lists = ldsc(lds_ld(&ws->mep->lists));
  const int size = nd.size;
  const int num_iter = c.encoder_speed == 0 ? 2 : 1;
  int min_ref0 = (J.frame_type == F_B && J.interp_ref > 0) ? 1 : 0, min_ref1 = 0;
  mv_t min0[4], min1[4];
  for (int i = 0; i < 4; i++) { min0[i] = mvp; min1[i] = mvp; }
  unsigned min_sad = 1u << 30;
  const PIX* oy = ws->org_y;
  const int osy = ws->org_sy;
  // a step whose inputs equal those of the previous step of the same list changes nothing (see bipred_par): skipped
  int prev_ref[2] = {-1, -1}, prev_cnt[2][kMaxRefs];
  mv_t prev_mv[2][4];
  for (int n = 0; n < num_iter; n++) {
    const int stop = part == 0 ? 0 : 1;
    for (int list = 1; list >= stop; list--) {
      mv_t mvo = list ? min0[0] : min1[0];
      int ref_o = list ? min_ref0 : min_ref1;
      {
        const mv_t* mo = list ? min0 : min1;
        int same = n > 0 && prev_ref[list] == ref_o;
        for (int i = 0; i < 4; i++) same = same && prev_mv[list][i].x == mo[i].x && prev_mv[list][i].y == mo[i].y;
        for (int r = 0; r < J.num_ref; r++) {
          const int cnt = lists->mvcand_num[r];
          same = same && prev_cnt[list][r] == cnt;
          prev_cnt[list][r] = cnt;
        }
        prev_ref[list] = ref_o;
        for (int i = 0; i < 4; i++) prev_mv[list][i] = mo[i];
        if (tk_uniform(same)) continue;
      }
      pred_inter_yuv<SP>(t, lds_ld(&J.ref[ref_o]), ws->pred_y, ws->pred_u, ws->pred_v, nd.ypos, nd.xpos, size, nd.bw, nd.bh,
                     list ? min0 : min1, J.sign[ref_o], c.width, c.height, c.enable_bipred, part > 0, c.bitdepth, 1);
      t.sync();
      build_org8<PIX, SP>(t, ws->org8, oy, osy, ws->pred_y, size, c.bitdepth);
      t.sync();
      int ref_start, ref_end;
      if (J.frame_type == F_P) { ref_start = 0; ref_end = J.num_ref - 1; }
      else {
        ref_start = ref_end = list ? 1 : 0;
        if (J.interp_ref) { ref_start++; ref_end++; }
      }
      for (int r = ref_start; r <= ref_end; r++) {
        mv_t mvp2 = (J.frame_type == F_B && list == 1) ? mvo : mvp;
        mv_t mv_all[4];
        unsigned sad = search_inter<PIX, SP>(t, J, ws, nd.ypos, nd.xpos, size, ws->org8, size, r, mv_center[r], mvp2, mv_all, part, J.sign[r]);
        add_cands4(t, ws, r, mv_all);
        if (sad < min_sad) {
          min_sad = sad;
          if (list) { min_ref1 = r; for (int i = 0; i < 4; i++) min1[i] = mv_all[i]; }
          else { min_ref0 = r; for (int i = 0; i < 4; i++) min0[i] = mv_all[i]; }
        }
      }
    }
  }
  *ref_idx0 = min_ref0;
  *ref_idx1 = min_ref1;
  for (int i = 0; i < 4; i++) { mv_arr0[i] = min0[i]; mv_arr1[i] = min1[i]; }
}

// search_intra_prediction_params (encode_block.c:928-1031): intra mode by luma SAD against the frame-edge
// prediction; evaluation order DC, HOR, VER, PLANAR (stop here when num_intra_modes == 4), then the six
// angular modes; first minimum wins.  DC is always built from (left, top) here (sic: `xposY >= 0` :953).
template <typename PIX, int SP>
TK_DEVNI unsigned intra_sad_search(const Team t, JobR<PIX> J, WsP<PIX> ws, const Node& nd, int num_modes, int* mode_out) {
  const auto& c = J.cfg;
  const int size = nd.size, bd = c.bitdepth;
  const int ur = upright_avail(nd.ypos, nd.xpos, size, size, c.width, kMaxSb);
  const int dl = downleft_avail(nd.ypos, nd.xpos, size, size, c.height, kMaxSb);
  const PIX* fy = J.rec.y + nd.ypos * J.rec.sy + nd.xpos;
  const PIX* oy = ws->org_y;
  const int osy = ws->org_sy;
  make_edges<SP>(t, ws->edgep, fy, J.rec.sy, (const PIX*)nullptr, 0, 0, 0, nd.ypos, nd.xpos, size, ur, dl, 0, bd);
  t.sync();
  unsigned min_sad = 1u << 30;
  int best = 0;
  const int n = num_modes == 4 ? 4 : 10;
  for (int e = 0; e < n; e++) {
    const int m = e == 0 ? 0 : e == 1 ? 2 : e == 2 ? 3 : e == 3 ? 1 : e;  // evaluation order -> intra_mode_t
    pred_intra<SP>(t, ws->edgep, 1, 1, size, ws->pred_y, size, m, bd);
    t.sync();
    int local = 0;
    for (int k = t.rank; k < size * size; k += t.size) {
      int i, j;
      split2(mk_pow2(size), k, i, j);
      local += iabs((int)spc<SP>(oy)[i * osy + j] - (int)spc<SP>(ws->pred_y)[k]);
    }
    const unsigned sad = (unsigned)team_sum(t, local) >> (bd - 8);
    t.sync();
    if (sad < min_sad) { min_sad = sad; best = m; }
  }
  *mode_out = best;
  return min_sad;
}

// ---------------------------------------------------------------------------------
// mode_decision_rdo (encode_block.c:1835-2121).  Result in nd.best; returns min cost.
// ---------------------------------------------------------------------------------
template <typename PIX, int SP>
TK_DEVNI unsigned mode_decision(const Team t, JobR<PIX> J, WsP<PIX> ws, Node& nd) {
  const auto& c = J.cfg;
  const int size = nd.size;
  const double lambda = J.lambda;
  const int rect = nd.bw != size || nd.bh != size;
  const int max_tb = c.enable_tb_split == 1 ? 2 : 1;
  const int max_pb = c.enable_pb_split ? 4 : 1;
  unsigned min_cost = kCostInit;
  int do_inter = 1, do_intra = 1;
  BlkParam p;
  // deterministic stand-in for the reference's uninitialised tmp_block_param
  p.mode = M_SKIP; p.intra_mode = 0; p.skip_idx = 0; p.pb_part = P_NONE; p.ref0 = p.ref1 = 0; p.dir = 0;
  p.tb_param = 0; p.tb_split = 0; p.cbp_y = p.cbp_u = p.cbp_v = 0;
  for (int i = 0; i < 4; i++) { p.mv0[i] = mk_mv(0, 0); p.mv1[i] = mk_mv(0, 0); }

  if (J.frame_type != F_I) {
    p.tb_param = 0;
    p.pb_part = P_NONE;
    for (int k = 0; k < nd.syn.num_skip; k++) {
      set_cand(p, nd.skip[k], k, M_SKIP);
      unsigned cost = rdo_trial<PIX, SP>(t, J, ws, nd, p, lambda);
      if (cost < min_cost) { min_cost = cost; if (t.rank == 0) keep_best(nd, p); }
    }
  }
  if ((size < 128 || c.encoder_speed == 0) && !rect) {
    if (J.frame_type != F_I) {
      for (int k = 0; k < nd.syn.num_merge; k++) {
        set_cand(p, nd.merge[k], k, M_MERGE);
        for (int tb = 0; tb <= max_tb - 1; tb++) {
          p.tb_param = (int8_t)tb;
          unsigned cost = rdo_trial<PIX, SP>(t, J, ws, nd, p, lambda, tb > 0, min_cost);
          if (cost < min_cost) { min_cost = cost; if (t.rank == 0) keep_best(nd, p); }
        }
      }
      // encoder_speed > 0: intra-vs-inter pre-decision by SAD (encode_block.c:1943-1947, 1990-1993)
      const int intra_inter_sad = c.encoder_speed > 0;
      unsigned sad_intra = 0xffffffffu;
      if (intra_inter_sad) {
        int im;
        sad_intra = intra_sad_search<PIX, SP>(t, J, ws, nd, J.num_intra_modes, &im);
        sad_intra += (unsigned)(int)mul_add_nofma(J.sqrt_lambda, 2.0, 0.5);
      }
      // uni-prediction per reference
      mv_t mv_center[kMaxRefs];
      mv_t mv_all[4][4];
      mv_t mvp = mk_mv(0, 0);
      const PIX* oy = ws->org_y;
      int min_idx = 0, max_idx = J.num_ref - 1;
      {
        const int br = ws->mep->lists->best_ref;
        if (!(br < 0 || c.encoder_speed < 2 || c.enable_bipred)) min_idx = max_idx = br;
      }
      if (J.frame_type == F_B && J.interp_ref > 2) min_idx = 1;
      unsigned worst_cost = 0, best_cost = 0xffffffffu;
      for (int r = min_idx; r <= max_idx; r++) {
        mvp = get_mv_pred(J.cells, J.cell_stride, nd.ypos, nd.xpos, c.width, c.height, size, kMaxSb);
        if (t.rank == 0) add_mvcand(ws->mep, r, mvp);
        t.sync();
        nd.syn.mvp = mvp;
        mv_center[r] = mvp;
        unsigned sad_inter = 0xffffffffu;
        for (int part = 0; part < max_pb; part++) {
          unsigned sad = search_inter<PIX, SP>(t, J, ws, nd.ypos, nd.xpos, size, oy, ws->org_sy, r, mv_center[r], mvp, mv_all[part], part, J.sign[r]);
          add_cands4(t, ws, r, mv_all[part]);
          mv_center[r] = mv_all[0][0];
          sad_inter = sad < sad_inter ? sad : sad_inter;
        }
        if (intra_inter_sad) {
          do_inter = sad_inter < sad_intra;
          if (sad_inter < sad_intra) do_intra = 0;
        }
        if (!do_inter) continue;
        p.mode = M_INTER;
        p.ref0 = p.ref1 = (int8_t)r;
        for (int part = 0; part < max_pb; part++) {
          p.pb_part = (int8_t)part;
          for (int i = 0; i < 4; i++) { p.mv0[i] = mv_all[part][i]; p.mv1[i] = mv_all[part][i]; }
          const int min_tb = c.encoder_speed < 1 ? -1 : 0;
          for (int tb = min_tb; tb <= max_tb - 1; tb++) {
            p.tb_param = (int8_t)tb;
            // worst/best cost feed only the encoder_speed 2 reference shortcut; where that is inactive the
            // exact costs of losing trials are never used and the trial may be pruned
            unsigned cost = rdo_trial<PIX, SP>(t, J, ws, nd, p, lambda, tb > min_tb, (c.encoder_speed < 2 || c.enable_bipred) ? min_cost : 0xffffffffu);
            worst_cost = cost > worst_cost ? cost : worst_cost;
            best_cost = cost < best_cost ? cost : best_cost;
            if (cost < min_cost) { min_cost = cost; if (t.rank == 0) keep_best(nd, p); }
          }
        }
      }
      // "one reference convincingly better": remember reference 0 for the rest of the SB (sic: best_ref_idx
      // is never updated in the reference, encode_block.c:1868/2018-2019); uint32 wrap-around as in C.
      if (worst_cost && worst_cost * 3u > best_cost * 4u) {
        t.sync();
        if (t.rank == 0) ws->mep->lists->best_ref = 0;
        t.sync();
      }
      // bi-prediction
      if (J.num_ref > 1 && c.enable_bipred && do_inter) {
        int r0, r1;
        mv_t a0[4], a1[4];
        search_bipred<PIX, SP>(t, J, ws, nd, 0, mv_center, mvp, &r0, &r1, a0, a1);
        p.mode = M_BIPRED;
        p.pb_part = P_NONE;
        p.ref0 = (int8_t)r0; p.ref1 = (int8_t)r1;
        for (int i = 0; i < 4; i++) { p.mv0[i] = a0[i]; p.mv1[i] = a1[i]; }
        for (int tb = 0; tb <= max_tb - 1; tb++) {
          p.tb_param = (int8_t)tb;
          unsigned cost = rdo_trial<PIX, SP>(t, J, ws, nd, p, lambda, tb > 0, min_cost);
          if (cost < min_cost) { min_cost = cost; if (t.rank == 0) keep_best(nd, p); }
        }
        if (J.frame_type == F_B && c.encoder_speed == 0) {
          // joint +mv / -mv search (search_bipred_prediction_params me_mode 1, encode_block.c:1708-1737, 2052-2068)
          const int ri0 = J.interp_ref ? 1 : 0, ri1 = J.interp_ref ? 2 : 1;
          const Plane3<PIX> f0 = lds_ld(&J.ref[ri0]);
          const Plane3<PIX> f1 = lds_ld(&J.ref[ri1]);
          MeArgs a;
          a.cb_size = size; a.ostride = ws->org_sy; a.width = size; a.height = size; a.rstride = f0.sy; a.sign = 0;
          a.fwidth = c.width; a.fheight = c.height; a.xpos = nd.xpos; a.ypos = nd.ypos; a.enable_bipred = 1;
          a.bitdepth = c.bitdepth; a.lam = J.sqrt_lambda; a.speed = c.encoder_speed; a.pu_x = nd.xpos; a.pu_y = nd.ypos;
          mv_t mvb;
          motion_estimate_bi<PIX, SP>(t, ws->mep, oy, f0.y + nd.ypos * f0.sy + nd.xpos, f1.y + nd.ypos * f1.sy + nd.xpos, a, mv_center[ri0],
                             mvp, ri0, &mvb);
          p.mode = M_BIPRED;
          p.pb_part = P_NONE;
          p.ref0 = (int8_t)ri0; p.ref1 = (int8_t)ri1;
          for (int i = 0; i < 4; i++) { p.mv0[i] = mvb; p.mv1[i] = mvb; }
          p.tb_param = 0;
          unsigned cost = rdo_trial<PIX, SP>(t, J, ws, nd, p, lambda);
          if (cost < min_cost) { min_cost = cost; if (t.rank == 0) keep_best(nd, p); }
        }
      }
    }
    // intra (encode_block.c:2070-2114).  The reference re-encodes the winning mode for both
    // tb_param values after the search; those trials are repeats of trials already made (same inputs,
    // deterministic), so their costs are taken from the search instead of being recomputed.
    p.mode = M_INTRA;
    int intra_mode = 0;
    unsigned best_tb_cost[2] = {kCostInit, kCostInit};
    if (!do_intra) {
    } else if (c.intra_rdo) {
      unsigned min_intra = kCostInit;
      for (int m = 0; m < J.num_intra_modes; m++) {
        p.intra_mode = (int8_t)m;
        unsigned tbc[2] = {kCostInit, kCostInit};
        int improved = 0;
        for (int tb = 0; tb <= max_tb - 1; tb++) {
          p.tb_param = (int8_t)tb;
          // only a cost below both the best intra cost and the best overall cost can change the outcome
          unsigned cost = rdo_trial<PIX, SP>(t, J, ws, nd, p, lambda, 0, min_intra < min_cost ? min_intra : min_cost);
          tbc[tb] = cost;
          if (cost < min_intra) { min_intra = cost; intra_mode = m; improved = 1; }
        }
        if (improved) { best_tb_cost[0] = tbc[0]; best_tb_cost[1] = tbc[1]; }
      }
      p.intra_mode = (int8_t)intra_mode;
      for (int tb = 0; tb <= max_tb - 1; tb++) {
        p.tb_param = (int8_t)tb;
        unsigned cost = best_tb_cost[tb];
        if (cost < min_cost) { min_cost = cost; p.cbp_y = p.cbp_u = p.cbp_v = 0; if (t.rank == 0) keep_best(nd, p); }
      }
    } else {
      intra_sad_search<PIX, SP>(t, J, ws, nd, J.num_intra_modes, &intra_mode);
      p.intra_mode = (int8_t)intra_mode;
      for (int tb = 0; tb <= max_tb - 1; tb++) {
        p.tb_param = (int8_t)tb;
        unsigned cost = rdo_trial<PIX, SP>(t, J, ws, nd, p, lambda);
        if (cost < min_cost) { min_cost = cost; if (t.rank == 0) keep_best(nd, p); }
      }
    }
  }
  return min_cost;
}

// ---------------------------------------------------------------------------------
// mode_decision_rdo for the encoder_speed 0 operating points, spread over the wavefronts of the workgroup.
//
// The reference walks the trials of a block one after the other and keeps the first strictly smaller cost
// (encode_block.c:1885-2114).  Given the entry state most of them are independent:
//   * the uni-prediction search of each reference (its candidate list mvcand[r] is private to the reference) and the
//     12 RDO trials that follow it,
//   * the skip / merge candidates,
//   * the 10 intra modes x 2 transform splits,
// and only the bi-prediction search needs something from the others (the PART_NONE vector of every reference).
// Every such unit is a work item in a queue in LDS; the waves pop items until the queue is empty; the wave that
// finishes the LAST reference search runs the bi-prediction item.  Every trial has the index of its position in the
// reference's evaluation order, the winner is the trial with the smallest (cost << 32 | order) - the very trial the
// sequential strict-'<' scan would keep - and pruning compares lower-bound keys with the shared minimum (PruneCtx).
// With a single wave the queue is simply processed in the reference's order.
// Evaluation order indices: skip k: k | merge k, tb: 2+2k+tb | inter r, part, tb: 6+12r+3part+(tb+1) |
// bipred tb: 54+tb, joint (B frames): 56 | intra m, tb: 57+2m+tb.
// ---------------------------------------------------------------------------------
TK_DEV BlkParam blank_param() {
  BlkParam p;
  p.mode = M_SKIP; p.intra_mode = 0; p.skip_idx = 0; p.pb_part = P_NONE; p.ref0 = p.ref1 = 0; p.dir = 0;
  p.tb_param = 0; p.tb_split = 0; p.cbp_y = p.cbp_u = p.cbp_v = 0;
  for (int i = 0; i < 4; i++) { p.mv0[i] = mk_mv(0, 0); p.mv1[i] = mk_mv(0, 0); }
  return p;
}

// Keep the trial that is in ws->rec_* / ws->coef_* (reconstruction and quantised coefficients of a square block of `size`)
// in the wave's snapshot buffers.
template <typename PIX, int SP>
TK_DEV void snapshot_trial(const Team t, WsP<PIX> ws, int size, const BlkParam& p) {
  BigWs<PIX>* g = ws->big;
  const int sc = size >> 1;
  copy_block<SP_GLOBAL, SP>(t, g->best_y, size, ws->rec_y, size, size, size);
  copy_block<SP_GLOBAL, SP>(t, g->best_u, sc, ws->rec_u, sc, sc, sc);
  copy_block<SP_GLOBAL, SP>(t, g->best_v, sc, ws->rec_v, sc, sc, sc);
  if (TKU(p.cbp_y) | TKU(p.cbp_u) | TKU(p.cbp_v)) {
    const int tbs = TKU(p.tb_split);
    const int qy = tbs ? tmin(size >> 1, (int)kMaxQuant) : tmin(size, (int)kMaxQuant);
    const int ny = (tbs ? 4 : 1) * qy * qy;
    const int csplit = tbs && sc > 4;
    const int qc = csplit ? tmin(sc >> 1, (int)kMaxQuant) : tmin(sc, (int)kMaxQuant);
    const int nc = (csplit ? 4 : 1) * qc * qc;
    const int16_t *cy = ws->coef_y, *cu = ws->coef_u, *cv = ws->coef_v;
    TK_GLOBAL int16_t* dy = gptr(g->best_cy);
    TK_GLOBAL int16_t* du = gptr(g->best_cu);
    TK_GLOBAL int16_t* dv = gptr(g->best_cv);
    for (int k = t.rank; k < ny; k += t.size) dy[k] = cy[k];
    for (int k = t.rank; k < nc; k += t.size) { du[k] = cu[k]; dv[k] = cv[k]; }
  }
}

template <typename PIX> struct MdCtx {
  Wg wg;
  WgShared* sh;
  Node* nd;
  unsigned long long mykey;  // best key among this wave's trials
};

// Returns 0 when the trial was dropped before it touched the prediction buffers (rdo_trial: `untouched`), 1 otherwise.
template <typename PIX, int SP>
TK_DEV int par_trial(const Team t, JobR<PIX> J, WsP<PIX> ws, MdCtx<PIX>& M, BlkParam& p, unsigned order, int reuse_pred) {
  int untouched = 0;
  const unsigned cost = rdo_trial<PIX, SP>(t, J, ws, *M.nd, p, J.lambda, reuse_pred, 0xffffffffu, &M.sh->bestkey, order, nullptr, &untouched);
  if (cost == (unsigned)kCostInit) return !untouched;  // pruned: cannot have the smallest key
  const unsigned long long key = ((unsigned long long)cost << 32) | order;
  if (key < M.mykey) {
    M.mykey = key;
    if (t.rank == 0) {
      lds_st(&M.sh->wbest[M.wg.wave], normalize_best(*M.nd, p));
      *ldsc(&M.sh->wkey[M.wg.wave]) = key;
      wg_min64(&M.sh->bestkey, key);
    }
    snapshot_trial<PIX, SP>(t, ws, tk_uniform(ldsc(M.nd)->size), p);
    t.sync();
  }
  return 1;
}

// Wait (with the limit of every intra-workgroup wait) until *p, an LDS word another wave releases, reaches `at_least`.
template <typename PIX> TK_DEV void wg_wait_at_least(const Team t, WsP<PIX> ws, int* p, int at_least) {
  (void)ws;
  const unsigned long long w0 = wg_clock();
  for (unsigned spins = 1;; spins++) {
    const int d = wg_load_acquire(p);   // every lane acquires (one broadcast LDS read)
    if (team_bcast0(t, d) >= at_least) break;
    if ((spins & 1023u) == 0 && tk_uniform64(wg_clock() - w0) > (unsigned long long)kWgWaitLimit) wg_wait_failed();
    wg_pause();
  }
}
// Telescope of the joint +mv / -mv search of a B frame (search_bipred_prediction_params me_mode 1, encode_block.c:1708-1737): claimed
// by the first wave that gets to it; result in sh->bj_sad / bj_mv, bj_state = 2.  Returns 0 when another wave has claimed it.
template <typename PIX, int SP>
TK_DEVNI int md_bijoint_telescope(const Team t, JobR<PIX> J, WsP<PIX> ws, MdCtx<PIX>& M) {
  const auto& c = J.cfg;
  WgShared* const sh_ = tk_uniform_ptr(M.sh);
  int mine = 0;
  if (t.rank == 0) mine = wg_cas(&sh_->bj_state, 0, 1);
  if (!team_bcast0(t, mine)) return 0;
  const auto ndl = ldsc(M.nd);
  struct { int size, ypos, xpos; } nd = {TKU(ndl->size), TKU(ndl->ypos), TKU(ndl->xpos)};
  const int ri0 = J.interp_ref ? 1 : 0, ri1 = J.interp_ref ? 2 : 1;
  const Plane3<PIX> f0 = lds_ld(&J.ref[ri0]);
  const Plane3<PIX> f1 = lds_ld(&J.ref[ri1]);
  MeArgs a;
  a.cb_size = nd.size; a.ostride = ws->org_sy; a.width = nd.size; a.height = nd.size; a.rstride = f0.sy; a.sign = 0;
  a.fwidth = c.width; a.fheight = c.height; a.xpos = nd.xpos; a.ypos = nd.ypos; a.enable_bipred = 1;
  a.bitdepth = c.bitdepth; a.lam = J.sqrt_lambda; a.speed = c.encoder_speed; a.pu_x = nd.xpos; a.pu_y = nd.ypos;
  mv_t mvb = mk_mv(0, 0);
  const mv_t ctr = lds_ld(&sh_->ref_mv[ri0][0][0]);   // PART_NONE vector of the first reference = mv_center[ri0]
  const unsigned sad = motion_estimate_bi<PIX, SP>(t, ws->mep, ws->org_y, f0.y + nd.ypos * f0.sy + nd.xpos, f1.y + nd.ypos * f1.sy + nd.xpos, a, ctr,
                                                   lds_ld(&sh_->mvp), ri0, &mvb, 1);
  t.sync();
  if (t.rank == 0) {
    *ldsc(&sh_->bj_sad) = sad;
    lds_st(&sh_->bj_mv, mvb);
    wg_store_release(&sh_->bj_state, 2);
  }
  t.sync();
  return 1;
}

template <typename PIX, int SP>
TK_DEVNI void md_item_bipred(const Team t, JobR<PIX> J, WsP<PIX> ws, MdCtx<PIX>& M) {
  const auto& c = J.cfg;
  Node& nd_ = *M.nd;
  const auto ndl = ldsc(&nd_);
  struct { int size, ypos, xpos; } nd = {TKU(ndl->size), TKU(ndl->ypos), TKU(ndl->xpos)};
  const int size = nd.size;
  const int max_tb = c.enable_tb_split == 1 ? 2 : 1;
  const mv_t mvp = lds_ld(&M.sh->mvp);
  mv_t mv_center[kMaxRefs];
  for (int r = 0; r < kMaxRefs; r++) mv_center[r] = lds_ld(&M.sh->mv_center[r]);
  int r0, r1;
  mv_t a0[4], a1[4];
  search_bipred<PIX, SP>(t, J, ws, nd_, 0, mv_center, mvp, &r0, &r1, a0, a1);
  BlkParam p = blank_param();
  p.mode = M_BIPRED; p.pb_part = P_NONE;
  p.ref0 = (int8_t)r0; p.ref1 = (int8_t)r1;
  for (int i = 0; i < 4; i++) { p.mv0[i] = a0[i]; p.mv1[i] = a1[i]; }
  for (int tb = 0, have_pred = 0; tb <= max_tb - 1; tb++) {
    p.tb_param = (int8_t)tb;
    have_pred |= par_trial<PIX, SP>(t, J, ws, M, p, 54u + (unsigned)tb, have_pred);
  }
  if (J.frame_type == F_B) {
    // joint +mv / -mv search (search_bipred_prediction_params me_mode 1, encode_block.c:1708-1737, 2052-2068)
    const int ri0 = J.interp_ref ? 1 : 0, ri1 = J.interp_ref ? 2 : 1;
    const Plane3<PIX> f0 = lds_ld(&J.ref[ri0]);
    const Plane3<PIX> f1 = lds_ld(&J.ref[ri1]);
    const PIX* oy = ws->org_y;
    MeArgs a;
    a.cb_size = size; a.ostride = ws->org_sy; a.width = size; a.height = size; a.rstride = f0.sy; a.sign = 0;
    a.fwidth = c.width; a.fheight = c.height; a.xpos = nd.xpos; a.ypos = nd.ypos; a.enable_bipred = 1;
    a.bitdepth = c.bitdepth; a.lam = J.sqrt_lambda; a.speed = c.encoder_speed; a.pu_x = nd.xpos; a.pu_y = nd.ypos;
    // the telescope of the joint search has run (or is running) on the wave that took the MD_BIJOINT item - or runs here if nobody
    // has claimed it yet; the extra candidates read the candidate list as the searches above left it
    WgShared* const sh_ = tk_uniform_ptr(M.sh);
    if (!md_bijoint_telescope<PIX, SP>(t, J, ws, M)) wg_wait_at_least(t, ws, &sh_->bj_state, 2);
    mv_t mvb = lds_ld(&sh_->bj_mv);
    const unsigned sad0 = (unsigned)tk_uniform((int)*ldsc(&sh_->bj_sad));
    motion_estimate_bi<PIX, SP>(t, ws->mep, oy, f0.y + nd.ypos * f0.sy + nd.xpos, f1.y + nd.ypos * f1.sy + nd.xpos, a, mv_center[ri0], mvp, ri0, &mvb, 2, sad0);
    p.mode = M_BIPRED; p.pb_part = P_NONE;
    p.ref0 = (int8_t)ri0; p.ref1 = (int8_t)ri1;
    for (int i = 0; i < 4; i++) { p.mv0[i] = mvb; p.mv1[i] = mvb; }
    p.tb_param = 0;
    par_trial<PIX, SP>(t, J, ws, M, p, 56u, 0);
  }
}

// MD_REF: the motion searches of one reference (encode_block.c:1966-1984) - partition after partition, each one seeded by
// the candidates the earlier ones left in mvcand[r].  The vectors of a partition are published as soon as it is searched.
template <typename PIX, int SP>
TK_DEVNI void md_item_ref(const Team t, JobR<PIX> J, WsP<PIX> ws, MdCtx<PIX>& M, int r) {
  const auto& c = J.cfg;
  const auto ndl = ldsc(M.nd);
  struct { int size, ypos, xpos; } nd = {TKU(ndl->size), TKU(ndl->ypos), TKU(ndl->xpos)};
  const int size = nd.size;
  const int max_pb = c.enable_pb_split ? 4 : 1;
  const mv_t mvp = lds_ld(&M.sh->mvp);
  const PIX* oy = ws->org_y;
  if (t.rank == 0) add_mvcand(ws->mep, r, mvp);
  t.sync();
  mv_t mv_center = mvp;
  if (t.rank == 0) ldsc(ws->mep)->cwin_valid = 0;
  t.sync();
  for (int part = 0; part < max_pb; part++) {
    mv_t mv_all[4];
    search_inter<PIX, SP>(t, J, ws, nd.ypos, nd.xpos, size, oy, ws->org_sy, r, mv_center, mvp, mv_all, part, J.sign[r]);
    add_cands4(t, ws, r, mv_all);
    if (part == 0) {
      mv_center = mv_all[0];
      // the eight searches of the HOR / VER / QUAD partitions all start from mv_center: one window for the block
      if (max_pb > 1) {
        const Plane3<PIX> rp = lds_ld(&J.ref[r]);
        me_stage_cb_window<PIX>(t, ws->mep, rp.y + nd.ypos * rp.sy + nd.xpos, rp.sy, nd.xpos, nd.ypos, size, mv_center, J.sign[r], c.width, c.height, r);
      }
    }
    if (t.rank == 0) {
      for (int i = 0; i < 4; i++) lds_st(&M.sh->ref_mv[r][part][i], mv_all[i]);
      if (part == max_pb - 1) lds_st(&M.sh->mv_center[r], mv_center);
      wg_fetch_add(&M.sh->parts_done[r], 1);   // release: the vectors above are visible to the wave that sees the count
    }
    t.sync();
  }
  if (t.rank == 0) ldsc(ws->mep)->cwin_valid = 0;   // the transform workspace the window lives in is about to be used again
  t.sync();
}

// MD_TRIAL: the RDO trials of one (reference, partition) (encode_block.c:1993-2012): tb_param -1 (no residual), 0 and 1 share one prediction.
template <typename PIX, int SP>
TK_DEVNI void md_item_trial(const Team t, JobR<PIX> J, WsP<PIX> ws, MdCtx<PIX>& M, int r, int part) {
  const auto& c = J.cfg;
  const int max_tb = c.enable_tb_split == 1 ? 2 : 1;
  r = tk_uniform(r); part = tk_uniform(part);
  WgShared* const sh_ = tk_uniform_ptr(M.sh);
  // the reference's search item was taken from the queue before this one: it is finished or running on another wave.  A wait of
  // kWgWaitLimit wall-clock ticks (a search item takes milliseconds) is a protocol error: wg_wait_failed() stops the kernel / the
  // simulation loudly instead of hanging the GPU (the host reports the aborted launch, thor_hip.cpp:run_superblocks).
  const unsigned long long w0 = wg_clock();
  TK_PROFMD_MARK(pwt_);
  for (unsigned spins = 1;; spins++) {
    const int d = wg_load_acquire(&sh_->parts_done[r]);   // every lane acquires (one broadcast LDS read)
    if (team_bcast0(t, d) > part) break;
    if ((spins & 1023u) == 0 && tk_uniform64(wg_clock() - w0) > (unsigned long long)kWgWaitLimit) wg_wait_failed();
    wg_pause();
  }
  if (TK_PROFMD_ON(2)) TK_PROFMD_ACC(ws, 20, pwt_);
  mv_t mv_all[4][4];
  for (int q = 0; q <= part; q++)
    for (int i = 0; i < 4; i++) mv_all[q][i] = lds_ld(&sh_->ref_mv[r][q][i]);
  // With enable_pb_split every inter trial predicts the four quadrants with mv0[0..3] whatever the partition, so a
  // partition whose quadrant vectors equal those of an EARLIER partition has the same prediction, residual, SSD and
  // coefficient bits as that one and strictly more header bits (longer partition code, more vector differences): its
  // cost is not smaller and its evaluation order is later - it can never be selected.  Skipped (exact).
  int dup = 0;
  for (int q = 0; q < part && c.enable_pb_split; q++) {
    int eq = 1;
    for (int i = 0; i < 4; i++) eq = eq && mv_all[q][i].x == mv_all[part][i].x && mv_all[q][i].y == mv_all[part][i].y;
    dup = dup || eq;
  }
  if (tk_uniform(dup)) return;
  BlkParam p = blank_param();
  p.mode = M_INTER;
  p.ref0 = p.ref1 = (int8_t)r;
  p.pb_part = (int8_t)part;
  for (int i = 0; i < 4; i++) { p.mv0[i] = mv_all[part][i]; p.mv1[i] = mv_all[part][i]; }
  for (int tb = -1, have_pred = 0; tb <= max_tb - 1; tb++) {
    p.tb_param = (int8_t)tb;
    have_pred |= par_trial<PIX, SP>(t, J, ws, M, p, 6u + 12u * (unsigned)r + 3u * (unsigned)part + (unsigned)(tb + 1), have_pred);
  }
}

// search_bipred_prediction_params (me_mode 0, PART_NONE) of a P frame with all waves in lock step.  The reference walks
// 2 iterations x {list 1, list 0}; inside one such step it searches EVERY reference against 2*org - pred of the other
// list and keeps the first strictly smaller SAD (encode_block.c:1770-1816).  The searches of one step are independent
// (each touches only its own candidate list), so wave w takes reference w; the leader (wave 0) builds 2*org - pred before
// and reduces in reference order after each step - the very scan of the reference.  Then the two trials (tb 0 / 1).
template <typename PIX, int SP>
TK_DEVNI void bipred_par(const Wg wg, const Team t, JobR<PIX> J, WsP<PIX> ws, MdCtx<PIX>& M) {
  const auto& c = J.cfg;
  WgShared* sh_ = M.sh;
  const auto sh = ldsc(sh_);
  const auto ndl = ldsc(M.nd);
  struct { int size, ypos, xpos, bw, bh; } nd = {TKU(ndl->size), TKU(ndl->ypos), TKU(ndl->xpos), TKU(ndl->bw), TKU(ndl->bh)};
  const auto lists = ldsc(lds_ld(&ws->mep->lists));
  const int size = nd.size;
  const int num_iter = c.encoder_speed == 0 ? 2 : 1;
  const int max_tb = c.enable_tb_split == 1 ? 2 : 1;
  const mv_t mvp = lds_ld(&sh_->mvp);
  // Round 5: NO leader phase.  Every wave keeps the state of the search (the two lists' best vectors / references, the running minimum) in its
  // own registers - the reduction after a step reads the four waves' results and is the same deterministic scan in every wave - and every wave
  // builds ITS QUARTER of the rows of 2*org - pred straight into the shared buffer (the leader used to predict and subtract the whole block while
  // three waves waited: a third of a step).  Two workgroup barriers per step as before: after the build, after the searches.
  // A step whose inputs - reference and vector of the other list, hence 2*org - pred; and the candidate list of every
  // reference - equal those of the previous step of the same list finds the same SADs again, none of which is below
  // min_sad any more (the earlier step left min_sad <= all of them): it changes nothing and is skipped (about a third of
  // all steps on typical content).  Exact, not a heuristic.
  int prev_ref[2] = {-1, -1}, prev_cnt[2][kMaxRefs];
  mv_t prev_mv[2][4];
  mv_t min0[4], min1[4];
  for (int i = 0; i < 4; i++) { min0[i] = mvp; min1[i] = mvp; }
  int ref0 = 0, ref1 = 0;
  unsigned min_sad = 1u << 30;
  if (wg.wave == 0 && t.rank == 0) sh->bp_org8 = ws->org8;   // the shared 2*org - pred block: wave 0's buffer (visible after the fork barrier? no: published below)
  t.sync();
  wg.barrier();
  PIX* const org8 = (PIX*)sh->bp_org8;   // same address space as this wave's buffers (same block size)
  const int whole = nd.bw == size && nd.bh == size;
  for (int n = 0; n < num_iter; n++)
    for (int list = 1; list >= 0; list--) {
      const int buf = (2 * n + (1 - list)) & 1;   // result slots alternate: a fast wave's next step never overwrites what a slow one still reads
      const mv_t* mo = list ? min0 : min1;
      const int ref_o = list ? ref0 : ref1;
      int same = n > 0 && prev_ref[list] == ref_o;
      for (int i = 0; i < 4; i++) same = same && prev_mv[list][i].x == mo[i].x && prev_mv[list][i].y == mo[i].y;
      for (int r = 0; r < J.num_ref; r++) {
        const int cnt = lists->mvcand_num[r];
        same = same && prev_cnt[list][r] == cnt;
        prev_cnt[list][r] = cnt;
      }
      prev_ref[list] = ref_o;
      for (int i = 0; i < 4; i++) prev_mv[list][i] = mo[i];
      if (tk_uniform(same)) continue;   // the same decision in every wave: all of them read the same counts and hold the same state
      if (whole) {
        // this wave's rows of 2*org - pred (get_inter_prediction_luma of the other list's vector, inter_prediction.c:93-183; encode_block.c:1786-1791)
        const Plane3<PIX> rp = lds_ld(&J.ref[ref_o]);
        const int sgn = J.sign[ref_o];
        const mv_t mvc_ = clip_mv(mo[0], nd.ypos, nd.xpos, c.width, c.height, size, size, sgn);
        const SubPel sp = luma_setup(mvc_, sgn, size, size, c.width, c.height, nd.xpos, nd.ypos, c.enable_bipred);
        const PIX* ry = rp.y + nd.ypos * rp.sy + nd.xpos;
        const int rows = (size + wg.nwaves - 1) / wg.nwaves, r0 = tmin(size, wg.wave * rows), r1 = tmin(size, r0 + rows);   // every row has an owner for any wave count
        const auto o8 = spc<SP>(org8);
        const auto oys = spc<SP>(ws->org_y);
        const int osy = ws->org_sy;
        const Pow2 pw = mk_pow2(size);
        for (int k = r0 * size + t.rank; k < r1 * size; k += t.size) {
          int i, j;
          split2(pw, k, i, j);
          o8[k] = (PIX)sat_pix(2 * (int)oys[i * osy + j] - luma_sample(ry, rp.sy, i, j, sp, c.enable_bipred, c.bitdepth), c.bitdepth);
        }
      } else if (wg.wave == 0) {   // blocks cut by the frame edge: the whole-block path on one wave
        pred_inter_yuv<SP>(t, lds_ld(&J.ref[ref_o]), ws->pred_y, ws->pred_u, ws->pred_v, nd.ypos, nd.xpos, size, nd.bw, nd.bh, mo, J.sign[ref_o], c.width,
                           c.height, c.enable_bipred, 0, c.bitdepth, 1);
        t.sync();
        build_org8<PIX, SP>(t, org8, ws->org_y, ws->org_sy, ws->pred_y, size, c.bitdepth);
      }
      t.sync();
      wg.barrier();   // 2*org - pred is complete
      for (int r = wg.wave; r < J.num_ref; r += wg.nwaves) {
        mv_t mv_all[4];
        const unsigned sad = search_inter<PIX, SP>(t, J, ws, nd.ypos, nd.xpos, size, org8, size, r, lds_ld(&sh_->mv_center[r]), mvp, mv_all, 0, J.sign[r]);
        add_cands4(t, ws, r, mv_all);
        if (t.rank == 0) { sh->bp_sad2[buf][r] = sad; for (int i = 0; i < 4; i++) lds_st(&sh_->bp_mv2[buf][r][i], mv_all[i]); }
      }
      t.sync();
      wg.barrier();   // every reference's result is there
      for (int r = 0; r < J.num_ref; r++) {   // the reference's scan (encode_block.c:1770-1816), in every wave
        const unsigned sd = (unsigned)tk_uniform((int)sh->bp_sad2[buf][r]);
        if (sd < min_sad) {
          min_sad = sd;
          if (list) { ref1 = r; for (int i = 0; i < 4; i++) min1[i] = lds_ld(&sh_->bp_mv2[buf][r][i]); }
          else { ref0 = r; for (int i = 0; i < 4; i++) min0[i] = lds_ld(&sh_->bp_mv2[buf][r][i]); }
        }
      }
    }
  // trials: tb 0 on wave 0, tb 1 on the next wave (each builds its own prediction)
  for (int tb = 0; tb <= max_tb - 1; tb++)
    if (wg.wave == tb % wg.nwaves) {
      BlkParam p = blank_param();
      p.mode = M_BIPRED; p.pb_part = P_NONE;
      p.ref0 = (int8_t)ref0; p.ref1 = (int8_t)ref1;
      for (int i = 0; i < 4; i++) { p.mv0[i] = min0[i]; p.mv1[i] = min1[i]; }
      p.tb_param = (int8_t)tb;
      par_trial<PIX, SP>(t, J, ws, M, p, 54u + (unsigned)tb, 0);
    }
}

// Executed by every wave of the workgroup between the fork and the join barrier.
template <typename PIX, int SP>
TK_MDW void md_worker_sp(const Wg wg, const Team t, JobR<PIX> J, WsP<PIX> ws) {
  WgShared* sh_ = ws->sh;
  const auto sh = ldsc(sh_);
  MdCtx<PIX> M;
  M.wg = wg; M.sh = sh_; M.nd = &sh_->stack[sh->node]; M.mykey = ~0ull;
  if (t.rank == 0) sh->wsnap[wg.wave] = (void*)ws->big;
  const auto ndl = ldsc(M.nd);
  ws_select(ws, tk_uniform(ndl->size));
  org_select(t, J, ws, tk_uniform(ndl->size), ndl->ypos, ndl->xpos, ndl->bw, ndl->bh, 0);
  const auto& c = J.cfg;
  const int max_tb = c.enable_tb_split == 1 ? 2 : 1;
  const int n_items = sh->n_items;
#if defined(THOR_PROF_MD) && defined(THOR_PROF) && !TK_HOST
  long long pmd_acc_[4] = {0, 0, 0, 0};
#endif
  for (;;) {
    int i = 0;
    if (t.rank == 0) i = wg_fetch_add(&sh_->next_item, 1);
    i = team_bcast0(t, i);
    if (i >= n_items) break;
    const int kind = team_bcast0(t, sh->items[i].kind), ia = team_bcast0(t, sh->items[i].a), ib = team_bcast0(t, sh->items[i].b);
    TK_PROFMD_MARK(pk_);
    if (kind == MD_SKIP) {
      BlkParam p = blank_param();
      set_cand(p, lds_ld(&M.nd->skip[ia]), ia, M_SKIP);
      par_trial<PIX, SP>(t, J, ws, M, p, (unsigned)ia, 0);
    } else if (kind == MD_MERGE) {
      BlkParam p = blank_param();
      set_cand(p, lds_ld(&M.nd->merge[ia]), ia, M_MERGE);
      for (int tb = 0, have_pred = 0; tb <= max_tb - 1; tb++) {
        p.tb_param = (int8_t)tb;
        have_pred |= par_trial<PIX, SP>(t, J, ws, M, p, 2u + 2u * (unsigned)ia + (unsigned)tb, have_pred);
      }
    } else if (kind == MD_INTRA) {
      BlkParam p = blank_param();
      p.mode = M_INTRA; p.intra_mode = (int8_t)ia; p.tb_param = (int8_t)ib;
      par_trial<PIX, SP>(t, J, ws, M, p, 57u + 2u * (unsigned)ia + (unsigned)ib, 0);
    } else if (kind == MD_REF) {
      md_item_ref<PIX, SP>(t, J, ws, M, ia);
      t.sync();
      int done = 0;
      if (t.rank == 0) done = wg_fetch_add(&sh_->refs_done, 1) + 1;
      done = team_bcast0(t, done);
      if (done == sh->n_ref_items && sh->do_bipred == 1) {
        md_item_bipred<PIX, SP>(t, J, ws, M);
      }
    } else if (kind == MD_TRIAL) {
      md_item_trial<PIX, SP>(t, J, ws, M, ia, ib);
    } else if (kind == MD_BIJOINT) {
      // its first reference's search item left the queue earlier: the PART_NONE vector is there or on its way
      wg_wait_at_least(t, ws, &sh_->parts_done[J.interp_ref ? 1 : 0], 1);
      md_bijoint_telescope<PIX, SP>(t, J, ws, M);
    }
#if defined(THOR_PROF_MD) && defined(THOR_PROF) && !TK_HOST
    if (TK_PROFMD_ON(1)) {
      const int slot_ = (kind == MD_SKIP || kind == MD_MERGE) ? 0 : kind == MD_INTRA ? 1 : (kind == MD_REF || kind == MD_BIJOINT) ? 2 : 3;
      if (TK_PROFMD_ON(8)) { const long long d_ = TK_CYC() - pk_; pmd_acc_[0] += slot_ == 0 ? d_ : 0; pmd_acc_[1] += slot_ == 1 ? d_ : 0; pmd_acc_[2] += slot_ == 2 ? d_ : 0; pmd_acc_[3] += slot_ == 3 ? d_ : 0; }
      else TK_PROF_ACC(ws, 16 + slot_, pk_);
    }
#endif
  }
#if defined(THOR_PROF_MD) && defined(THOR_PROF) && !TK_HOST
  if (TK_PROFMD_ON(8) && t.rank == 0) for (int q_ = 0; q_ < 4; q_++) ws->prof[16 + q_] += pmd_acc_[q_];
#endif
  if (tk_uniform(sh->do_bipred) == 2) {  // uniform over the workgroup: every wave takes part (same number of barriers); a scalar branch - no
                                         // workgroup barrier behind an exec-masked one (scripts/check_barrier_hazard.py)
    t.sync();
    TK_PROF_MARK(pb_);
    wg.barrier();            // every reference search has finished: mv_center[] and the candidate lists are final
    bipred_par<PIX, SP>(wg, t, J, ws, M);
    TK_PROF_ACC(ws, 27, pb_);
  }
}
// The decision code exists twice: for coding blocks whose sample buffers live in LDS (up to kLdsBlk) and for the larger ones
// (global scratch); see tk_common.h SP_LDS / SP_GLOBAL.
template <typename PIX>
TK_DEV void md_worker(const Wg wg, const Team t, JobR<PIX> J, WsP<PIX> ws) {
  const auto sh = ldsc(ws->sh);
  const int size = tk_uniform(ldsc(&ws->sh->stack[sh->node])->size);
  if (size <= kLdsBlk) md_worker_sp<PIX, SP_LDS>(wg, t, J, ws);
  else md_worker_sp<PIX, SP_GLOBAL>(wg, t, J, ws);
}

// Parked waves: woken by the master at every fork until it posts WG_CMD_EXIT at the end of the superblock.
template <typename PIX>
TK_DEV void wg_helper_loop(const Wg wg, const Team t, JobR<PIX> J, WsP<PIX> ws) {
  for (;;) {
    TK_PROF_MARK(ph_);
    wg.barrier();
    TK_PROF_ACC(ws, 28, ph_);   // parked while the master works alone (quadtree walk, early skip, final encodes)
    const int cmd = team_bcast0(t, ws->sh->cmd);
    if (cmd == WG_CMD_EXIT) { wg.barrier(); break; }  // second barrier: every wave has read the command before the master reuses it
#ifdef THOR_PROF
    { TK_PROF_MARK(pw_); md_worker(wg, t, J, ws); TK_PROF_ACC(ws, 5, pw_); }
#else
    md_worker(wg, t, J, ws);
#endif
    t.sync();
    wg.barrier();
  }
}

// Master side.  Result in nd.best; returns min cost.
template <typename PIX>
TK_MDW unsigned mode_decision_par(const Wg wg, const Team t, JobR<PIX> J, WsP<PIX> ws, int node, int* win_wave) {
  const auto& c = J.cfg;
  WgShared* sh_ = ws->sh;
  const auto sh = ldsc(sh_);
  const auto nd = ldsc(&sh_->stack[node]);
  const int max_tb = c.enable_tb_split == 1 ? 2 : 1;
  const int inter = J.frame_type != F_I;
  TK_PROFMD_MARK(pqs_);
  mv_t mvp = mk_mv(0, 0);
  if (inter) mvp = get_mv_pred(J.cells, J.cell_stride, nd->ypos, nd->xpos, c.width, c.height, nd->size, kMaxSb);
  t.sync();
  if (t.rank == 0) {
    int n = 0;
    auto push = [&](int kind, int a, int b) { sh->items[n].kind = (int8_t)kind; sh->items[n].a = (int8_t)a; sh->items[n].b = (int8_t)b; sh->items[n].pad = 0; n++; };
    static_assert(2 + 2 + kMaxRefs + 1 + 2 * kNumIntraModes + 4 * kMaxRefs <= kMdMaxItems, "work queue too small");
    static_assert(6 + 12 * kMaxRefs <= 54, "evaluation-order layout: the reference trials must end before the bi-prediction trials");
    if (inter) {
      for (int k = 0; k < nd->syn.num_skip; k++) push(MD_SKIP, k, 0);
      for (int k = 0; k < nd->syn.num_merge; k++) push(MD_MERGE, k, 0);
      for (int r = 0; r < J.num_ref; r++) push(MD_REF, r, 0);
      if (J.num_ref > 1 && c.enable_bipred && J.frame_type == F_B) push(MD_BIJOINT, 0, 0);   // B frames: telescope of the joint search
      nd->syn.mvp.x = mvp.x; nd->syn.mvp.y = mvp.y;
    }
    for (int m = 0; m < J.num_intra_modes; m++)
      for (int tb = 0; tb <= max_tb - 1; tb++) push(MD_INTRA, m, tb);
    // the trials of the searched vectors come last: by the time the queue gets here most searches have published theirs
    if (inter)
      for (int part = 0; part < (c.enable_pb_split ? 4 : 1); part++)
        for (int r = 0; r < J.num_ref; r++) { push(MD_TRIAL, r, part); sh->parts_done[r] = 0; }
    sh->n_items = n; sh->next_item = 0;
    sh->refs_done = 0; sh->n_ref_items = inter ? J.num_ref : 0;
    sh->do_bipred = (inter && J.num_ref > 1 && c.enable_bipred) ? (J.frame_type == F_P ? 2 : 1) : 0;
    sh->bj_state = 0;
    sh->node = node; lds_st(&sh_->mvp, mvp);
    sh->bestkey = ~0ull;
    for (int w = 0; w < kWaves; w++) sh->wkey[w] = ~0ull;
    sh->cmd = WG_CMD_MD;
  }
  t.sync();
  if (TK_PROFMD_ON(4)) TK_PROFMD_ACC(ws, 21, pqs_);
  wg.barrier();   // fork
#ifdef THOR_PROF
  { TK_PROF_MARK(pw_); md_worker(wg, t, J, ws); TK_PROF_ACC(ws, 5, pw_); t.sync(); wg.barrier(); TK_PROF_ACC(ws, 29, pw_); }
#else
  md_worker(wg, t, J, ws);
  t.sync();
  wg.barrier();   // join
#endif
  unsigned long long best = ~0ull;
  int bw = 0;
  for (int w = 0; w < wg.nwaves; w++) {
    const unsigned long long k = sh->wkey[w];
    if (k < best) { best = k; bw = w; }
  }
  if (best == ~0ull) return kCostInit;
  if (t.rank == 0) lds_st(&sh_->stack[node].best, lds_ld(&sh_->wbest[bw]));
  t.sync();
  *win_wave = bw;
  return (unsigned)(best >> 32);
}

// ---------------------------------------------------------------------------------
// Early skip (encode_block.c:2123-2392)
// ---------------------------------------------------------------------------------
template <typename PIX, int SP>
TK_DEV int early_skip_sub(const Team t, JobR<PIX> J, WsP<PIX> ws, const PIX* org_, int ostride,
                          const PIX* pred_, int pstride, int size, int qp, float thr) {
  const auto org = spc<SP>(org_);
  const auto pred = spc<SP>(pred_);
  const auto xin = ldsc(ws->xfp->in);
  const auto xcoef = ldsc(ws->xfp->coef);
  // luma: 2x2 average + (N/2) transform (size > 4 always here), threshold 0.5*thr
  const int bd = J.cfg.bitdepth;
  const int s2 = size / 2;
  for (int k = t.rank; k < s2 * s2; k += t.size) {
    int i, j;
    split2(mk_pow2(s2), k, i, j);
    int a = (int16_t)((int)org[(2 * i) * ostride + 2 * j] - (int)pred[(2 * i) * pstride + 2 * j]);
    int b = (int16_t)((int)org[(2 * i) * ostride + 2 * j + 1] - (int)pred[(2 * i) * pstride + 2 * j + 1]);
    int cc = (int16_t)((int)org[(2 * i + 1) * ostride + 2 * j] - (int)pred[(2 * i + 1) * pstride + 2 * j]);
    int d = (int16_t)((int)org[(2 * i + 1) * ostride + 2 * j + 1] - (int)pred[(2 * i + 1) * pstride + 2 * j + 1]);
    xin[i * s2 + j] = (int16_t)((a + b + cc + d + 2) >> 2);  // row-major (fwd_core layout)
  }
  t.sync();
  fwd_transform_block(t, ws->xfp, s2, bd);
  const int shift2 = 21 - ilog2(s2) + qp / 6;
  const double fql = (double)(1 << shift2) / (double)quant_scale(qp % 6);
  const double rel = 0.5 * thr;  // float -> double promotion as in the reference
  const int threshold = (int)(rel * fql);
  int f = 0;
  for (int k = t.rank; k < s2 * s2; k += t.size)
    if (iabs((int)xcoef[k]) > threshold) f = 1;
  const int r = team_ballot(t, f) != 0ull;   // any lane: no LDS flag round trip
  t.sync();
  return r;
}

template <typename PIX, int SP>
TK_DEV int early_skip_subC(const Team t, JobR<PIX> J, WsP<PIX> ws, const PIX* org_, int ostride,
                           const PIX* pred_, int pstride, int size, int qp, float thr) {
  const auto org = spc<SP>(org_);
  const auto pred = spc<SP>(pred_);
  const int shift2 = 21 - 5 + qp / 6;
  const double fql = (double)(1 << shift2) / (double)quant_scale(qp % 6);
  const int threshold = ((int)(thr * fql)) << (J.cfg.bitdepth - 8);
  // calc_cbp as the reference EXECUTES it, i.e. calc_cbp_simd (enc/enc_kernels.c:827-907, selected at
  // encode_block.c:2225 because use_simd = 1): int16 column sums of the residual; for 16/8 wide
  // blocks |sum| > thr per column; for 4x4 the SIMD code tests (col[2k+1] + |col[2k]|) > thr, which
  // is NOT the scalar |col[2k] + col[2k+1]| > thr - the oracle binary runs the SIMD form.
  const int ncol = size == 4 ? 2 : size;
  int f = 0;
  for (int col = t.rank; col < ncol; col += t.size) {
    if (size == 4) {
      int lo = 0, hi = 0;
      for (int i = 0; i < 4; i++) {
        lo = (int16_t)(lo + (int16_t)((int)org[i * ostride + 2 * col] - (int)pred[i * pstride + 2 * col]));
        hi = (int16_t)(hi + (int16_t)((int)org[i * ostride + 2 * col + 1] - (int)pred[i * pstride + 2 * col + 1]));
      }
      if (hi + (int)(int16_t)iabs(lo) > threshold) f = 1;
    } else {
      int sum = 0;
      for (int i = 0; i < size; i++) sum = (int16_t)(sum + (int16_t)((int)org[i * ostride + col] - (int)pred[i * pstride + col]));
      if ((int16_t)iabs(sum) > (int16_t)threshold) f = 1;
    }
  }
  const int r = team_ballot(t, f) != 0ull;
  t.sync();
  return r;
}

template <typename PIX, int SP>
TK_DEVNI int check_early_skip(const Team t, JobR<PIX> J, WsP<PIX> ws, const Node& nd, const BlkParam& p) {
  const auto& c = J.cfg;
  const int size = nd.size, size0 = size < 32 ? size : 32;
  const int qpY = J.qp, qpC = TK_TAB.chroma_qp[qpY];
  float thr = c.early_skip_thr;
  if (c.encoder_speed > 1 && nd.size == kMaxSb) thr += thr / 4;  // encode_block.c:2256-2257
  const int size0c = size0 >> 1;
  int significant = 0;
  for (int i = 0; i < size && !significant; i += size0)
    for (int j = 0; j < size && !significant; j += size0) {
      struct { int ypos, xpos; } sub = {nd.ypos + i, nd.xpos + j};
      const int yc = sub.ypos >> 1, xc = sub.xpos >> 1;
      if (p.dir == 2) {
        pred_inter_yuv<SP>(t, lds_ld(&J.ref[p.ref0]), ws->p0_y, ws->p0_u, ws->p0_v, sub.ypos, sub.xpos, size0, size0, size0, p.mv0,
                       J.sign_ge[p.ref0], c.width, c.height, c.enable_bipred, 0, c.bitdepth);
        pred_inter_yuv<SP>(t, lds_ld(&J.ref[p.ref1]), ws->p1_y, ws->p1_u, ws->p1_v, sub.ypos, sub.xpos, size0, size0, size0, p.mv1,
                       J.sign_ge[p.ref1], c.width, c.height, c.enable_bipred, 0, c.bitdepth);
        t.sync();
        average_yuv<SP>(t, ws->pred_y, ws->pred_u, ws->pred_v, ws->p0_y, ws->p0_u, ws->p0_v, ws->p1_y, ws->p1_u, ws->p1_v,
                    size0, size0, size0);
      } else {
        pred_inter_yuv<SP>(t, lds_ld(&J.ref[p.ref0]), ws->pred_y, ws->pred_u, ws->pred_v, sub.ypos, sub.xpos, size0, size0, size0,
                       p.mv0, J.sign[p.ref0], c.width, c.height, c.enable_bipred, 0, c.bitdepth);
      }
      t.sync();
      significant = early_skip_sub<PIX, SP>(t, J, ws, ws->org_y + i * ws->org_sy + j, ws->org_sy, ws->pred_y, size0,
                                   size0, qpY, thr);
      if (!significant)
        significant = early_skip_subC<PIX, SP>(t, J, ws, ws->org_u + (i >> 1) * ws->org_sc + (j >> 1), ws->org_sc, ws->pred_u, size0c, size0c, qpC, thr);
      if (!significant)
        significant = early_skip_subC<PIX, SP>(t, J, ws, ws->org_v + (i >> 1) * ws->org_sc + (j >> 1), ws->org_sc, ws->pred_v, size0c, size0c, qpC, thr);
    }
  return !significant;
}

// ---------------------------------------------------------------------------------
// Final encode of a CB: recompute (encode_block final), write recon + cell state, emit bits.
// ---------------------------------------------------------------------------------
template <typename PIX, int SP>
TK_DEVNI int final_encode(const Team t, JobR<PIX> J, WsP<PIX> ws, Node& nd, BitSink& out, const BigWs<PIX>* snap = nullptr,
                          int trial_bits = -1) {
  TK_PROF_T0();
  TK_PROFMD_MARK(pfe0_);
  BlkParam p = lds_ld(&nd.best);
  const int size = nd.size, sc = size >> 1;
  const int yc = nd.ypos >> 1, xc = nd.xpos >> 1;
  int nbits = 0;
  if (snap) {
    // the winning trial of the parallel decision left its reconstruction and coefficients in its wave's snapshot: emit + copy
    {
      // cooperative emission: every lane runs the syntax on the same values, lane 0 writes the words (tk_bits.h:bs_coeff_team)
      BitSink w = out;
      w.store = t.rank == 0;
      bs_open(w);
      bs_block_t<true>(w, lds_ld(&nd.syn), p, snap->best_cy, snap->best_cu, snap->best_cv, &t, nullptr);
      bs_close(w);
      out.ovf |= w.ovf;
      nbits = w.pos - out.pos;
    }
    nbits = team_bcast0(t, nbits);
    out.pos += nbits;
    if (TK_PROFMD_ON(4)) TK_PROFMD_ACC(ws, 24, pfe0_);
    copy_block<SP_GLOBAL, SP_GLOBAL>(t, J.rec.y + nd.ypos * J.rec.sy + nd.xpos, J.rec.sy, snap->best_y, size, nd.bw, nd.bh);
    copy_block<SP_GLOBAL, SP_GLOBAL>(t, J.rec.u + yc * J.rec.sc + xc, J.rec.sc, snap->best_u, sc, nd.bw >> 1, nd.bh >> 1);
    copy_block<SP_GLOBAL, SP_GLOBAL>(t, J.rec.v + yc * J.rec.sc + xc, J.rec.sc, snap->best_v, sc, nd.bw >> 1, nd.bh >> 1);
  } else {
    if (trial_bits >= 0) {
      // skip candidate whose trial was the block's last encode_block: its reconstruction is still in ws->rec_*, nd.best carries
      // the fields the trial set, a skip block has no coefficients - the second encode_block would rebuild the same state
      nbits = trial_bits;
    } else {
      BitSink cnt;
      cnt.buf = nullptr; cnt.pos = 0; cnt.cap = 0; cnt.emit = 0; cnt.ovf = 0;
      nbits = encode_block<PIX, SP>(t, J, ws, nd, p, cnt);
    }
    // bits (cooperative emission), then recon copy and cells
    {
      BitSink w = out;
      w.store = t.rank == 0;
      bs_open(w);
      bs_block_t<true>(w, lds_ld(&nd.syn), p, ws->coef_y, ws->coef_u, ws->coef_v, &t, nullptr);
      bs_close(w);
      out.ovf |= w.ovf;
    }
    out.pos += nbits;
    copy_block<SP_GLOBAL, SP>(t, J.rec.y + nd.ypos * J.rec.sy + nd.xpos, J.rec.sy, ws->rec_y, size, nd.bw, nd.bh);
    copy_block<SP_GLOBAL, SP>(t, J.rec.u + yc * J.rec.sc + xc, J.rec.sc, ws->rec_u, sc, nd.bw >> 1, nd.bh >> 1);
    copy_block<SP_GLOBAL, SP>(t, J.rec.v + yc * J.rec.sc + xc, J.rec.sc, ws->rec_v, sc, nd.bw >> 1, nd.bh >> 1);
  }
  // copy_deblock_data (encode_block.c:1568-1613)
  const int tbs = p.tb_param > 0 ? 1 : 0;
  const int pb = p.mode == M_INTER ? p.pb_part : P_NONE;
  const int div = size / (2 * kMinPb);
  const int cw = nd.bw / kMinPb, ch = nd.bh / kMinPb;
  const int cbpbits = tbs ? 7 : ((p.cbp_y ? 1 : 0) | (p.cbp_u ? 2 : 0) | (p.cbp_v ? 4 : 0));
  for (int k = t.rank; k < cw * ch; k += t.size) {
    int m, n;
    split2(mk_div(cw), k, m, n);
    int m0 = div > 0 ? m / div : 0, n0 = div > 0 ? n / div : 0;
    int index = 2 * m0 + n0;
    DbCell& cell = J.cells[(nd.ypos / kMinPb + m) * J.cell_stride + nd.xpos / kMinPb + n];
    cell.mv0 = p.mv0[index];
    cell.mv1 = p.mv1[index];
    cell.mode = (uint8_t)p.mode;
    cell.size = (uint8_t)size;
    cell.tbpb = (uint8_t)(tbs | (pb << 1));
    cell.cbp = (uint8_t)cbpbits;
    cell.ref0 = p.ref0;
    cell.ref1 = p.ref1;
    cell.dir = p.dir;
    cell.pad = 0;
  }
  t.sync();
  TK_PROF_ADD(ws, PF_FINAL);
  return nbits;
}

// ---------------------------------------------------------------------------------
// process_block (encode_block.c:2401-2566) as an explicit-stack traversal of one superblock.
// ---------------------------------------------------------------------------------
// Runs on the master wave (wg.wave == 0); the other waves of the workgroup sit in wg_helper_loop meanwhile.  The
// shared tables (ws->sh->tabs) must have been filled (xform_tables_fill).
template <typename PIX>
TK_DEV void process_sb(const Wg wg, const Team t, JobR<PIX> J, WsP<PIX> ws, int sb_y, int sb_x, BitSink& out) {
  TK_PROF_T0();
  const auto& c = J.cfg;
  const int fw = c.width, fh = c.height;
  if (t.rank == 0)
  { MeLists* L = ws->mep->lists; for (int r = 0; r < kMaxRefs; r++) { L->mvcand_num[r] = 0; L->mvcand_mask[r] = 0; } L->best_ref = -1; }
  t.sync();
  if (J.stats && J.frame_type != F_I && t.rank == 0) {
    team_add64(&J.stats[2], 1ull);
    team_add64(&J.stats[3], (unsigned long long)(tmin((int)kMaxSb, fw - sb_x) * tmin((int)kMaxSb, fh - sb_y)));
  }
  int sp = 0;
  unsigned ret = 0;  // value "returned" by the node that was just popped
  int have_ret = 0;
  {
    Node& n = ws->stack[0];
    if (t.rank == 0) { n.size = kMaxSb; n.ypos = sb_y; n.xpos = sb_x; n.stage = 0; }
    t.sync();
  }
  while (sp >= 0) {
    Node& nd = ws->stack[sp];
    if (nd.stage == 0) {
      // ---- entry
      const int size = nd.size, ypos = nd.ypos, xpos = nd.xpos;
      if (ypos + kMinBlk > fh || xpos + kMinBlk > fw) { ret = 0; have_ret = 1; sp--; continue; }
      TK_PROFMD_MARK(pen_);
      ws_select(ws, tk_uniform(size));
      t.sync();
      if (t.rank == 0) {
        nd.bw = tmin(size, fw - xpos);
        nd.bh = tmin(size, fh - ypos);
        nd.encode_this_size = ypos + size <= fh && xpos + size <= fw;
        nd.encode_rect = !nd.encode_this_size && J.frame_type != F_I;
        nd.cost_small = 1u << 28;
        nd.bitpos0 = out.pos;
        nd.child = 0;
        nd.md_done = 0;
        nd.cost_this = 1u << 28;
        SynCtx& s = nd.syn;
        s.frame_type = J.frame_type; s.num_ref = J.num_ref; s.enable_bipred = c.enable_bipred; s.interp_ref = J.interp_ref;
        s.max_pb_part = c.enable_pb_split ? 4 : 1; s.max_tb_part = c.enable_tb_split == 1 ? 2 : 1;
        s.num_intra_modes = J.num_intra_modes; s.size = size; s.encode_this_size = nd.encode_this_size;
        s.num_skip = 0; s.num_merge = 0; s.mvp = mk_mv(0, 0);
        find_contexts(J.cells, J.cell_stride, ypos, xpos, fh, fw, size, c.use_block_contexts, &s);
        if (J.frame_type != F_I && (nd.encode_this_size || nd.encode_rect)) {
          s.num_skip = get_mv_cands(J.cells, J.cell_stride, ypos, xpos, fw, fh, size, kMaxSb, nd.skip);
          s.num_merge = get_mv_cands(J.cells, J.cell_stride, ypos, xpos, fw, fh, size, kMaxSb, nd.merge);
        }
      }
      t.sync();
      org_select(t, J, ws, tk_uniform(size), ypos, xpos, tk_uniform(nd.bw), tk_uniform(nd.bh), 1);
      if (TK_PROFMD_ON(4)) TK_PROFMD_ACC(ws, 22, pen_);
      TK_PROFMD_MARK(pes_);
      // ---- early skip
      const int lds_blk = tk_uniform(size <= kLdsBlk);   // address space of this block's sample buffers (SP_LDS / SP_GLOBAL instances)
      if (nd.encode_this_size && J.frame_type != F_I && c.early_skip_thr > 0.0f) {
        unsigned min_cost = kCostInit;
        int any = 0;
        int best_bits = -1;   // >= 0: the best candidate's trial was the last encode_block (its reconstruction is in ws->rec_*)
        BlkParam p;
        p.intra_mode = 0; p.pb_part = P_NONE; p.tb_param = 0; p.tb_split = 0; p.cbp_y = p.cbp_u = p.cbp_v = 0;
        for (int k = 0; k < nd.syn.num_skip; k++) {
          set_cand(p, nd.skip[k], k, M_SKIP);
          TK_PROF_T0();
          int es_ = lds_blk ? check_early_skip<PIX, SP_LDS>(t, J, ws, nd, p) : check_early_skip<PIX, SP_GLOBAL>(t, J, ws, nd, p);
          TK_PROF_ADD(ws, PF_ESKIP);
          if (es_) {
            any = 1;
            // The check of a block of up to 32x32 predicts the whole block in one piece with the arguments predict_inter uses for
            // a uni-directional skip candidate, into the same buffers, and only reads them afterwards: the trial takes it over.
            const int reuse = tk_uniform(size <= 32 && p.dir != 2) ? 2 : 0;
            int nb = 0;
            unsigned cost = lds_blk ? rdo_trial<PIX, SP_LDS>(t, J, ws, nd, p, J.lambda, reuse, 0xffffffffu, nullptr, 0, &nb)
                                    : rdo_trial<PIX, SP_GLOBAL>(t, J, ws, nd, p, J.lambda, reuse, 0xffffffffu, nullptr, 0, &nb);
            if (cost < min_cost) { min_cost = cost; best_bits = nb; if (t.rank == 0) keep_best(nd, p); t.sync(); }
            else best_bits = -1;
          }
        }
        if (any) {
          if (J.stats && t.rank == 0) {
            team_add64(&J.stats[0], (unsigned long long)(nd.bw * nd.bh));
            if (nd.size == kMaxSb) team_add64(&J.stats[1], 1ull);
          }
          const int nbits = lds_blk ? final_encode<PIX, SP_LDS>(t, J, ws, nd, out, nullptr, best_bits)
                                    : final_encode<PIX, SP_GLOBAL>(t, J, ws, nd, out, nullptr, best_bits);
          (void)nbits;
          // The reference recomputes cost_calc on the final reconstruction (encode_block.c:2483-2488): the final encode repeats the
          // winning trial (same prediction, same bits), so that value is the trial's cost.
          ret = min_cost;
#if TK_HOST
          if (getenv("THOR_DBG")) fprintf(stderr, "F %d y %d x %d s %d ES mode %d idx %d cost %u bits %d\n", J.frame_num, nd.ypos, nd.xpos, nd.size, nd.best.mode, nd.best.skip_idx, ret, nbits);
#endif
          have_ret = 1;
          sp--;
          if (TK_PROFMD_ON(4)) TK_PROFMD_ACC(ws, 23, pes_);
          continue;
        }
      }
      if (TK_PROFMD_ON(4)) TK_PROFMD_ACC(ws, 23, pes_);
      // ---- split signalling + children (bottom-up), unless this is a top-down 16x16 (encode_block.c:2418)
      const int top_down = size == 2 * kMinBlk && nd.encode_this_size && J.frame_type != F_I && c.encoder_speed > 0;
      if (size > kMinBlk && !top_down) {
        if (t.rank == 0) {
          BitSink w = out;
          bs_open(w);
          bs_super_mode(w, nd.syn, 0, 0, 1);
          bs_close(w);
          out.ovf |= w.ovf;
          nd.cost_small = 0;
        }
        {
          BitSink cnt = out;
          cnt.emit = 0;
          bs_super_mode(cnt, nd.syn, 0, 0, 1);
          out.pos = cnt.pos;
        }
        t.sync();
        if (t.rank == 0) nd.stage = 1;
        t.sync();
      } else {
        if (t.rank == 0) nd.stage = 2;
        t.sync();
      }
      have_ret = 0;
      continue;
    }
    if (nd.stage == 1) {
      // ---- children TL, BL, TR, BR (encode_block.c:2513-2516)
      if (have_ret) {
        if (t.rank == 0) nd.cost_small += ret;
        have_ret = 0;
        t.sync();
      }
      if (nd.child < 4) {
        const int ch = nd.child, hs = nd.size / 2;
        Node& cn = ws->stack[sp + 1];
        t.sync();
        if (t.rank == 0) {
          cn.size = hs;
          cn.ypos = nd.ypos + ((ch & 1) ? hs : 0);   // order: (0,0) (1,0) (0,1) (1,1) in (y,x)
          cn.xpos = nd.xpos + ((ch & 2) ? hs : 0);
          cn.stage = 0;
          nd.child = ch + 1;
        }
        t.sync();
        sp++;
        continue;
      }
      if (t.rank == 0) nd.stage = 2;
      t.sync();
    }
    // ---- stage 2: decide this size
    {
      ws_select(ws, tk_uniform(nd.size));
      org_select(t, J, ws, tk_uniform(nd.size), nd.ypos, nd.xpos, tk_uniform(nd.bw), tk_uniform(nd.bh), 1);  // the children replaced the LDS copy
      unsigned cost = 1u << 28;
      int snap_wave = -1;
      if (nd.encode_this_size || nd.encode_rect) {
        if (!nd.md_done) {
          const int rect = nd.bw != nd.size || nd.bh != nd.size;
          if (c.encoder_speed == 0 && c.intra_rdo && !rect) cost = mode_decision_par(wg, t, J, ws, sp, &snap_wave);
          else cost = nd.size <= kLdsBlk ? mode_decision<PIX, SP_LDS>(t, J, ws, nd) : mode_decision<PIX, SP_GLOBAL>(t, J, ws, nd);
          t.sync();
#if TK_HOST
          if (getenv("THOR_DBG")) fprintf(stderr, "F %d y %d x %d s %d RDO mode %d cost %u small %u ref %d part %d tb %d mv %d %d\n", J.frame_num, nd.ypos, nd.xpos, nd.size, nd.best.mode, cost, nd.cost_small, nd.best.ref0, nd.best.pb_part, nd.best.tb_param, nd.best.mv0[0].x, nd.best.mv0[0].y);
#endif
          // top-down 16x16 (encoder_speed > 0, encode_block.c:2418-2419, 2528-2537): the children are only
          // tried when this size costs more than size^2 * iq_8x8[qp] / 8
          const int top_down = nd.size == 2 * kMinBlk && nd.encode_this_size && J.frame_type != F_I && c.encoder_speed > 0;
          if (top_down && cost > (unsigned)(nd.size * nd.size * TK_TAB.iq_8x8[J.qp] / 8)) {
            out.pos = nd.bitpos0;
            if (t.rank == 0) {
              BitSink w = out;
              bs_open(w);
              bs_super_mode(w, nd.syn, 0, 0, 1);
              bs_close(w);
              out.ovf |= w.ovf;
              nd.cost_small = 0;
              nd.md_done = 1;
              nd.cost_this = cost;
              nd.stage = 1;
            }
            {
              BitSink cnt = out;
              cnt.emit = 0;
              bs_super_mode(cnt, nd.syn, 0, 0, 1);
              out.pos = cnt.pos;
            }
            t.sync();
            have_ret = 0;
            continue;
          }
        } else cost = nd.cost_this;
        if (cost <= nd.cost_small) {
          out.pos = nd.bitpos0;
          // decided just now by the parallel decision: the winner's wave still holds its trial (snapshot_trial)
          const BigWs<PIX>* snap = snap_wave >= 0 ? (const BigWs<PIX>*)ldsc(ws->sh)->wsnap[snap_wave] : nullptr;
          TK_PROFMD_MARK(pfe_);
          if (nd.size <= kLdsBlk) final_encode<PIX, SP_LDS>(t, J, ws, nd, out, snap);
          else final_encode<PIX, SP_GLOBAL>(t, J, ws, nd, out, snap);
          if (TK_PROFMD_ON(4)) TK_PROFMD_ACC(ws, 25, pfe_);   // whole call; slot 24 holds the emission part of the snapshot path
        }
      }
      ret = cost < nd.cost_small ? cost : nd.cost_small;
      have_ret = 1;
      sp--;
    }
  }
  // release the parked waves
  t.sync();
  if (t.rank == 0) ws->sh->cmd = WG_CMD_EXIT;
  t.sync();
  wg.barrier();
  wg.barrier();  // all parked waves have seen WG_CMD_EXIT (see wg_helper_loop)
  TK_PROF_ADD(ws, PF_SB);
}

}  // namespace tk
