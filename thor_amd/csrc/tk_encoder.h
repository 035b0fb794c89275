// tk_encoder.h - host side of the engine: stream state in device memory, per-frame job set-up,
// superblock wavefront scheduling, bitstream assembly.  Shared by the HIP library (kernels in
// thor_hip.cpp) and by the 1-lane host simulation used in CPU tests (hostsim.cpp); the two
// differ only in the `backend` functions declared below.
// Specification followed: enc/encode_frame.c:637-850 (encode_frame: lambda, header, SB raster
// order, deblock, CDEF, reference rotation), enc/write_bits.c:49-121 (sequence / frame header),
// enc/putbits.c:45-83 (frame framing), enc/mainenc.c:246-523 (low-delay GOP: frame types, QPs and
// reference lists).
#pragma once
#include <vector>
#include <string>
#include <thread>
#include <cstdio>
#include <cstring>
#include <cmath>
#include "tk_common.h"
#include "tk_tables.h"
#include "tk_cdef.h"
#include "tk_interp_dev.h"
#include "tk_clpf.h"

namespace tk {

// ---- backend (device memory + launches) ------------------------------------------------
namespace backend {
void* dev_alloc(size_t n);
void dev_free(void* p);
void h2d(void* d, const void* h, size_t n);
void d2h(void* h, const void* d, size_t n);
void dev_memset(void* d, int v, size_t n);
void dev_sync();
size_t team_ws_bytes(int pix_bytes);
// jobs: device array of S FrameJob; hjobs: the same on the host.  ranges (host array of S, or nullptr = whole frames): the anti-diagonals
// t = l + 2k of the superblock grid, [lo, hi), that this launch codes of every stream (tk_sched.h; lo == hi: nothing).
struct SbRange { unsigned short lo, hi; };
template <typename PIX> void run_superblocks(const FrameJob<PIX>* jobs, const FrameJob<PIX>* hjobs, int S, const SbRange* ranges);
template <typename PIX> void run_clpf_stats(const ClpfJob<PIX>* jobs, const ClpfJob<PIX>* hjobs, int S);
template <typename PIX> void run_clpf_apply(const ClpfJob<PIX>* jobs, const ClpfJob<PIX>* hjobs, int S);  // copies rec -> src first
template <typename PIX> void run_deblock(const FrameJob<PIX>* jobs, const FrameJob<PIX>* hjobs, int S);
template <typename PIX> void run_make_ref(const FrameJob<PIX>* hjobs, const Plane3<PIX>* dst, int S);
// copies rec -> src, then runs the five CDEF passes; cjobs/hcjobs: device/host arrays of S CdefJob
template <typename PIX> void run_cdef(const CdefJob<PIX>* cjobs, const CdefJob<PIX>* hcjobs, int S);
// bit-level concatenation of per-SB bit strings: item i copies nbits[i] bits from src[i] to bit offset dst_bit[i] of dst
struct GatherItem { const uint32_t* src; int nbits; long long dst_bit; };
void run_gather(const GatherItem* d_items, int n, uint32_t* dst);
void release_superblocks(const void* jobs);
// temporally interpolated reference frames of n streams (all phases: pyramid, block search per level, merge, upscale,
// motion compensation, padding); jobs/hjobs: device/host arrays of n idev::Job
template <typename PIX> void run_interp(const idev::Job<PIX>* jobs, const idev::Job<PIX>* hjobs, int n);  // frees the scheduler state run_superblocks keeps for this job array
}  // namespace backend

// ---- parameters -------------------------------------------------------------------------
struct SeqParams {  // the enc_params fields this path honours (enc/mainenc.h:35-112)
  int width = 0, height = 0, qp = 32;
  int bitdepth = 8, input_bitdepth = 8;
  float frame_rate = 30.f;
  float lambda_coeffI = 1.f, lambda_coeffP = 1.f, lambda_coeffB = 1.f, lambda_coeffB0 = 1.f, lambda_coeffB1 = 1.f, lambda_coeffB2 = 1.f, lambda_coeffB3 = 1.f;
  float early_skip_thr = 0.f;
  int enable_tb_split = 0, enable_pb_split = 0, max_num_ref = 1, HQperiod = 1;
  int num_reorder_pics = 0, dyadic_coding = 1, interp_ref = 0;
  int dqpP = 0, dqpI = 0, dqpB = 0, dqpB0 = 0, dqpB1 = 0, dqpB2 = 0, dqpB3 = 0;
  float mqpP = 1.f, mqpB = 1.f, mqpB0 = 1.f, mqpB1 = 1.f, mqpB2 = 1.f, mqpB3 = 1.f;
  int intra_period = 0, intra_rdo = 0, encoder_speed = 0;
  int deblocking = 1, cdef = 2, clpf = 0, max_clpf_strength = 4, use_block_contexts = 0, enable_bipred = 0;
  int cfl_intra = 1, cfl_inter = 0;
  int log2_sb_size = 7;
};

struct FrameParams {
  int frame_type = F_I, qp = 32, num_ref = 0, frame_num = 0, interp_ref = 0, num_intra_modes = 10;
  int ref_array[kMaxRefs] = {0, 0, 0, 0};  // indices into the sliding window (0 = most recent); -1 = interpolated frame
  int interp_src[2] = {0, 0};              // window indices the interpolated frame is built from (before pruning)
  double lambda_coeff = 1.0;
  int b_level = 0;
};

// Coding-order schedule of one stream: a resumable restatement of the frame loop of enc/mainenc.c:246-625
// (frame types, QPs, b_level, reference lists for low-delay and dyadic hierarchical-B GOPs, the fall-back
// to PPP coding for a tail that does not fill a sub-GOP, duplicate / pre-intra reference removal).
// frame numbers are relative to `skip` like encoder_info.frame_info.frame_num.
struct GopScheduler {
  SeqParams sp;             // HQperiod / num_reorder_pics mutate at the tail exactly like the reference's params
  int skip = 0, num_frames = 0, file_frames = 0;
  int frame_num0 = 0, k = 0, sub_gop = 1, num_encoded = 0, last_intra = 0, last_PorI = -1, min_interp_depth = 0;
  bool started = false;
  static int ilog2i(unsigned v) { int n = 0; while (v >>= 1) n++; return n; }
  static int code_to_display(int sub, int idx) {
    static const int cd1[1] = {0}, cd2[2] = {1, 0}, cd4[4] = {3, 1, 0, 2}, cd8[8] = {7, 3, 1, 5, 0, 2, 4, 6},
                     cd16[16] = {15, 7, 3, 11, 1, 5, 9, 13, 0, 2, 4, 6, 8, 10, 12, 14};
    const int* t[5] = {cd1, cd2, cd4, cd8, cd16};
    return t[ilog2i(sub)][idx];
  }
  static int display_to_code(int sub, int idx) {
    static const int dc1[2] = {-1, 0}, dc2[3] = {-2, 1, 0}, dc4[5] = {-4, 2, 1, 3, 0}, dc8[9] = {-8, 4, 2, 5, 1, 6, 3, 7, 0},
                     dc16[17] = {-16, 8, 4, 9, 2, 10, 5, 11, 1, 12, 6, 13, 3, 14, 7, 15, 0};
    const int* t[5] = {dc1, dc2, dc4, dc8, dc16};
    return t[ilog2i(sub)][idx];
  }
  void init(const SeqParams& p, int skip_, int nframes, int file_frames_) {
    sp = p; skip = skip_; num_frames = nframes; file_frames = file_frames_;
    sub_gop = p.num_reorder_pics + 1 > 1 ? p.num_reorder_pics + 1 : 1;
    min_interp_depth = ilog2i((unsigned)(p.num_reorder_pics + 1)) - 3;
    if (p.frame_rate > 30) min_interp_depth--;
    frame_num0 = skip; k = 0; num_encoded = 0; last_intra = 0; last_PorI = -1; started = true;
  }
  // ring_frame_num(idx): frame_num of the reconstruction currently at sliding-window position idx.
  // Returns false when the sequence is finished; otherwise fills f and the absolute input frame index.
  template <class RingF> bool next(FrameParams& f, int& abs_frame, RingF ring_frame_num) {
    for (;;) {
      if (!(frame_num0 < skip + num_frames && frame_num0 + 1 <= file_frames)) return false;
      if (k >= sub_gop) {
        // end of a sub-GOP: tail fall-back (mainenc.c:615-623)
        if ((frame_num0 + sub_gop + 1 > file_frames || frame_num0 + sub_gop >= skip + num_frames) && sub_gop >= 2) {
          sp.HQperiod = sub_gop; sub_gop = 1; sp.num_reorder_pics = 0;
        }
        frame_num0 += sub_gop;  // the for-increment runs after the fall-back, i.e. with the NEW sub_gop
        k = 0;
        continue;
      }
      const int dyadic = sp.dyadic_coding;
      int frame_offset;
      if (dyadic && sub_gop > 1) frame_offset = code_to_display(sub_gop, k) - sub_gop + 1;
      else frame_offset = k == 0 ? 0 : k - sub_gop;
      const int frame_num_abs = frame_num0 + frame_offset;
      k++;
      if (frame_num_abs < skip) continue;
      f = FrameParams();
      f.frame_num = frame_num_abs - skip;
      abs_frame = frame_num_abs;
      if (sp.num_reorder_pics == 0) {
        if (sp.intra_period > 0) f.frame_type = (num_encoded % sp.intra_period) == 0 ? F_I : F_P;
        else f.frame_type = num_encoded == 0 ? F_I : F_P;
      } else {
        if (sp.intra_period > 0) f.frame_type = (f.frame_num % sp.intra_period) == 0 ? F_I : ((f.frame_num % sub_gop) == 0 ? F_P : F_B);
        else f.frame_type = f.frame_num == 0 ? F_I : ((f.frame_num % sub_gop) == 0 ? F_P : F_B);
      }
      const int coded_phase = (num_encoded + sub_gop - 2) % sub_gop + 1;
      const int b_level = ilog2i((unsigned)coded_phase);
      f.b_level = b_level;
      if (f.frame_type == F_I) { f.qp = sp.qp + sp.dqpI; last_intra = f.frame_num; }
      else if (sp.num_reorder_pics == 0) {
        if (num_encoded % sp.HQperiod) f.qp = (int)(sp.mqpP * (float)sp.qp) + sp.dqpP; else f.qp = sp.qp;
      } else {
        if (f.frame_num % sub_gop) {
          if (dyadic) {
            if (b_level == 0) f.qp = (int)(sp.mqpB0 * (float)sp.qp) + sp.dqpB0;
            else if (b_level == 1) f.qp = (int)(sp.mqpB1 * (float)sp.qp) + sp.dqpB1;
            else if (b_level == 2) f.qp = (int)(sp.mqpB2 * (float)sp.qp) + sp.dqpB2;
            else if (b_level == 3) f.qp = (int)(sp.mqpB3 * (float)sp.qp) + sp.dqpB3;
            else f.qp = (int)(sp.mqpB * (float)sp.qp) + sp.dqpB;
          } else f.qp = (int)(sp.mqpB * (float)sp.qp) + sp.dqpB;
        } else {
          if (f.frame_num % sp.HQperiod) f.qp = (int)(sp.mqpP * (float)sp.qp) + sp.dqpP; else f.qp = sp.qp;
        }
      }
      f.qp = f.qp < 0 ? 0 : (f.qp > 51 ? 51 : f.qp);
      f.num_ref = f.frame_type == F_I ? 0 : (num_encoded < sp.max_num_ref ? num_encoded : sp.max_num_ref);
      f.interp_ref = 0;
      int ra[kMaxRefs + 2] = {0, 0, 0, 0, 0, 0};
      auto imin = [](int a, int b) { return a < b ? a : b; };
      if (f.num_ref > 0) {
        if (sp.num_reorder_pics > 0) {
          if (dyadic) {
            if ((num_encoded - 1) % sub_gop == 0) {  // P frame: previous anchors
              ra[0] = num_encoded == 1 ? 0 : sub_gop - 1;
              if (f.num_ref > 1) ra[1] = imin(32, imin(num_encoded - 1, 2 * sub_gop - 1));
              for (int r = 2; r < f.num_ref; r++) ra[r] = r - 2;
            } else {
              const int display_phase = (f.frame_num - 1) % sub_gop;
              const int ref_offset = sub_gop >> (b_level + 1);
              if (b_level >= min_interp_depth && sp.interp_ref == 1) {
                if (f.num_ref == 2) f.num_ref++;
                f.interp_ref = sp.interp_ref;
                ra[1] = imin(num_encoded - 1, coded_phase - display_to_code(sub_gop, display_phase - ref_offset + 1) - 1);
                ra[2] = imin(num_encoded - 1, coded_phase - display_to_code(sub_gop, display_phase + ref_offset + 1) - 1);
                ra[0] = -1;
                f.interp_src[0] = ra[1]; f.interp_src[1] = ra[2];
                for (int r = 3; r < f.num_ref; r++) ra[r] = r - 3;
              } else {
                ra[0] = imin(num_encoded - 1, coded_phase - display_to_code(sub_gop, display_phase - ref_offset + 1) - 1);
                ra[1] = imin(num_encoded - 1, coded_phase - display_to_code(sub_gop, display_phase + ref_offset + 1) - 1);
                for (int r = 2; r < f.num_ref; r++) ra[r] = r - 2;
              }
            }
          } else {
            fprintf(stderr, "thor_hip: non-dyadic reordering is not implemented\n");
            abort();
          }
        } else {
          ra[0] = last_PorI;
          const int r1 = ((num_encoded + sp.HQperiod - 2) % sp.HQperiod) + 1;
          const int r2 = r1 == 1 ? 2 : 1;
          int r3 = r2 + 1;
          if (r3 == r1) r3 += 1;
          if (f.num_ref == 2) ra[1] = r1;
          else if (f.num_ref == 3) { ra[1] = r1; ra[2] = r2; }
          else if (f.num_ref == 4) { ra[1] = r1; ra[2] = r2; ra[3] = r3; }
        }
      }
      for (int r = f.num_ref - 1; r > 0; --r)
        for (int q = r - 1; q >= 0; --q)
          if (ra[q] == ra[r]) {
            for (int s2 = r; s2 < f.num_ref - 1; ++s2) ra[s2] = ra[s2 + 1];
            f.num_ref--;
            break;
          }
      if (f.frame_num > last_intra)
        for (int r = f.num_ref - 1; r >= 0; --r)
          if (ra[r] >= 0 && ring_frame_num(ra[r]) < last_intra) {
            for (int s2 = r; s2 < f.num_ref - 1; ++s2) ra[s2] = ra[s2 + 1];
            f.num_ref--;
          }
      for (int r = 0; r < kMaxRefs; r++) f.ref_array[r] = ra[r];
      f.num_intra_modes = (sp.intra_rdo == 0 || (f.frame_type != F_I && sp.encoder_speed > 0)) ? 4 : 10;
      if (f.frame_type == F_I) f.lambda_coeff = sp.lambda_coeffI;
      else if (f.frame_type == F_P) f.lambda_coeff = sp.lambda_coeffP;
      else f.lambda_coeff = b_level == 0 ? sp.lambda_coeffB0 : b_level == 1 ? sp.lambda_coeffB1 : b_level == 2 ? sp.lambda_coeffB2
                                                                              : b_level == 3 ? sp.lambda_coeffB3 : sp.lambda_coeffB;
      return true;
    }
  }
  // call after the frame returned by next() has been encoded (mainenc.c:552, 611)
  void advance(const FrameParams& f) {
    num_encoded++;
    last_PorI = f.frame_type != F_B ? 0 : last_PorI + 1;
  }
};

// ---- host bit writer (MSB first) ---------------------------------------------------------
struct HostBits {
  std::vector<uint32_t> w;  // word i holds stream bits [32i, 32i+32), first bit in the MSB
  int nbits = 0;
  void put(int n, uint32_t v) {
    if (n <= 0) return;
    if (n < 32) v &= (1u << n) - 1u;
    const int off = nbits & 31, room = 32 - off;
    if (off == 0) w.push_back(0);
    if (n <= room) {
      w.back() |= v << (room - n);
    } else {
      const int lo = n - room;
      w.back() |= v >> lo;
      w.push_back(v << (32 - lo));
    }
    nbits += n;
  }
  // append nb bits taken MSB-first from 32-bit words (device BitSink layout)
  void append_words(const uint32_t* src, int nb) {
    int i = 0;
    for (; i + 32 <= nb; i += 32) put(32, src[i >> 5]);
    if (i < nb) put(nb - i, src[i >> 5] >> (32 - (nb - i)));
  }
  void overwrite(int pos, int n, uint32_t v) {  // header back-patching, bit by bit (tiny)
    for (int i = n - 1; i >= 0; i--, pos++) {
      const uint32_t m = 0x80000000u >> (pos & 31);
      if ((v >> i) & 1u) w[pos >> 5] |= m; else w[pos >> 5] &= ~m;
    }
  }
  int get(int pos) const { return (int)((w[pos >> 5] >> (31 - (pos & 31))) & 1u); }
  size_t num_bytes() const { return ((size_t)nbits + 7) / 8; }
  void to_bytes(std::vector<uint8_t>& out) const {  // big-endian words -> byte stream
    const size_t nb = num_bytes();
    for (size_t i = 0; i < nb; i++) out.push_back((uint8_t)(w[i >> 2] >> (24 - 8 * (i & 3))));
  }
  void clear() { w.clear(); nbits = 0; }
};

inline void write_sequence_header(HostBits& b, const SeqParams& p) {  // write_bits.c:49-81 (4:2:0, no qmtx)
  b.put(16, p.width); b.put(16, p.height); b.put(3, p.log2_sb_size);
  b.put(1, p.enable_pb_split); b.put(1, p.enable_tb_split); b.put(2, p.max_num_ref - 1);
  b.put(2, p.interp_ref); b.put(1, 0 /*delta qp*/); b.put(1, p.deblocking); b.put(1, p.clpf ? 1 : 0);
  b.put(1, p.use_block_contexts); b.put(2, p.enable_bipred); b.put(1, 0 /*qmtx*/);
  b.put(2, 1 /*420*/); b.put(4, p.num_reorder_pics); b.put(1, p.cfl_intra); b.put(1, p.cfl_inter);
  b.put(1, p.bitdepth != 8);
  if (p.bitdepth != 8) b.put(1, p.bitdepth == 12);
  b.put(1, p.input_bitdepth != 8);
  if (p.input_bitdepth != 8) b.put(1, p.input_bitdepth == 12);
}

struct CdefHeader {
  int damping = 5, bits = 0;
  int strengths[8] = {0}, uv_strengths[8] = {0};
};
inline void write_cdef_params(HostBits& b, int pos_or_minus1, int cdef_on, const CdefHeader& h, int phase = 0) {  // write_bits.c:83-97
  HostBits tmp;
  if (cdef_on) {
    tmp.put(2, h.damping - 3); tmp.put(2, h.bits);
    for (int i = 0; i < (1 << h.bits); i++) { tmp.put(7, h.strengths[i]); tmp.put(7, h.uv_strengths[i]); }
  } else tmp.put(18, 0);
  if (pos_or_minus1 < 0) {
    for (int i = 0; i < tmp.nbits; i++) b.put(1, (uint32_t)tmp.get(i));
  } else {
    // Back-patch as the reference's stream writer performs it (enc/putbits.c:130-144): write_stream_pos()
    // back to the saved header position, rewrite, write_stream_pos() forward again.  The forward move restores
    // the saved 32-bit accumulator of the CURRENT position, so patched bits that fall into the stream's last,
    // not yet flushed word are lost and the originally written bits stay (visible on tiny frames).
    // `phase`: bit position (mod 32) of b's bit 0 in the stream the reference would be writing to (0 unless the bits
    // are appended to a caller's stream that already holds unflushed bits - drop-in seam, frame 0 after the sequence header).
    const int unflushed = (((b.nbits + phase) >> 5) << 5) - phase;
    for (int i = 0; i < tmp.nbits; i++)
      if (pos_or_minus1 + i < unflushed) b.overwrite(pos_or_minus1 + i, 1, (uint32_t)tmp.get(i));
  }
}

// ---- per-stream device state ---------------------------------------------------------------
template <typename PIX> struct DevFrame {
  PIX* base_y = nullptr;
  PIX* base_c = nullptr;
  Plane3<PIX> p;
  int frame_num = -1;
  void alloc(int w, int h, int pad) {
    int sy = (w + 2 * pad + 15) & ~15, sc = (w / 2 + 2 * (pad / 2) + 15) & ~15;
    size_t ay = (size_t)(h + 2 * pad) * sy + 64, ac = (size_t)(h / 2 + 2 * (pad / 2)) * sc + 64;
    base_y = (PIX*)backend::dev_alloc(ay * sizeof(PIX));
    base_c = (PIX*)backend::dev_alloc(2 * ac * sizeof(PIX));
    p.sy = sy; p.sc = sc;
    p.y = base_y + (size_t)pad * sy + pad;
    p.u = base_c + (size_t)(pad / 2) * sc + pad / 2;
    p.v = p.u + ac;
  }
  void release() { backend::dev_free(base_y); backend::dev_free(base_c); base_y = base_c = nullptr; }
};

template <typename PIX> struct Stream {
  DevFrame<PIX> orig, rec, tmp, interp;
  std::vector<DevFrame<PIX>> ring;  // sliding window, ring[0] = most recent reconstruction
  DbCell* cells = nullptr;
  uint32_t* sb_bits = nullptr;
  int* sb_nbits = nullptr;
  int* sb_status = nullptr;
  uint8_t* scratch = nullptr;
  // CDEF state
  int8_t* cdef_dir = nullptr; int* cdef_var = nullptr; int* cdef_fbc = nullptr; unsigned long long* cdef_mse = nullptr;
  int* cdef_sel = nullptr; int* cdef_fbsel = nullptr; CdefResult* cdef_res = nullptr; unsigned long long* cdef_tot = nullptr;
  uint32_t* clpf_stats = nullptr; uint8_t* clpf_fb_on = nullptr;   // CLPF: per-8x8 statistics, per-filter-block switches
  // temporal interpolation (interp_ref = 1): pyramid planes and per-level vector fields, device resident
  int ip_levels = 0;
  PIX* ip_pic[2][idev::kMaxLevels] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};  // allocation bases
  int ip_stride[idev::kMaxLevels] = {0, 0, 0, 0};
  idev::imv* ip_mv[idev::kMaxLevels][5] = {};   // mv0, mv1, nmv0, nmv1, guide per level
  int* ip_prog[idev::kMaxLevels] = {nullptr, nullptr, nullptr, nullptr};   // progress counters + ticket (last entry)
  int num_encoded = 0;
  GopScheduler gop;          // coding-order schedule (initialised by begin_sequence or lazily as open-ended low delay)
  FrameParams cur;           // frame returned by the last schedule()
  int cur_abs = 0;           // its absolute input frame index
  HostBits bits;             // bits of the frame being assembled (sequence header rides on frame 0)
  int bit_phase = 0;         // raw_frames mode: bit position mod 32 of the caller's stream at frame start
  std::vector<uint8_t> out;  // finished stream bytes (4-byte big-endian length + payload per frame)
};

// The caller-visible fields of the reference's deblock_data_t (common/types.h:178-187) in declaration order, from one DbCell:
// cbp.y/u/v are 0/1 in the reference too (quantize() returns a flag; transform-split blocks store 1/1/1, encode_block.c:1495-1498).
struct DdFields { int mode, cbp_y, cbp_u, cbp_v, size, tb_split, pb_part, mv0x, mv0y, mv1x, mv1y, ref_idx0, ref_idx1, bipred_flag; };
static inline DdFields dd_fields(const DbCell& c) {
  DdFields d;
  d.mode = c.mode; d.cbp_y = c.cbp & 1; d.cbp_u = (c.cbp >> 1) & 1; d.cbp_v = (c.cbp >> 2) & 1;
  d.size = c.size; d.tb_split = c.tbpb & 1; d.pb_part = c.tbpb >> 1;
  d.mv0x = c.mv0.x; d.mv0y = c.mv0.y; d.mv1x = c.mv1.x; d.mv1y = c.mv1.y;
  d.ref_idx0 = c.ref0; d.ref_idx1 = c.ref1; d.bipred_flag = c.dir;
  return d;
}

template <typename PIX> class Engine {
 public:
  SeqParams sp;
  int S = 0, sb_cols = 0, sb_rows = 0, nsb = 0, max_diag = 0, ring_size = 0;
  static const int kSbWords = 16384;
  std::vector<Stream<PIX>> st;
  FrameJob<PIX>* d_jobs = nullptr;
  std::vector<FrameJob<PIX>> h_jobs;
  CdefJob<PIX>* d_cjobs = nullptr;
  std::vector<CdefJob<PIX>> h_cjobs;
  ClpfJob<PIX>* d_ljobs = nullptr;
  std::vector<ClpfJob<PIX>> h_ljobs;
  int nfb_h = 0, nfb_v = 0;
  size_t ws_bytes = 0;
  int* d_nbits_all = nullptr; int* d_status_all = nullptr;   // [S][nsb]
  uint32_t* d_payload = nullptr; size_t payload_words = 0;   // compacted SB bits of all streams
  backend::GatherItem* d_items = nullptr;
  bool external_interp = false;  // drop-in mode: the caller uploads st[s].interp itself
  bool raw_frames = false;  // drop-in mode: no sequence header / framing; caller consumes st[s].bits
  unsigned long long* d_stats = nullptr;  // FrameJob::stats (4 counters + spare)
  long long* d_prof = nullptr;  // 32 cycle counters summed over all superblocks (THOR_PROF builds)
  idev::Job<PIX>* d_ijobs = nullptr;
  std::vector<idev::Job<PIX>> h_ijobs;

  void open(const SeqParams& p, int num_streams) {
    sp = p; S = num_streams;
    sb_cols = (p.width + kMaxSb - 1) / kMaxSb; sb_rows = (p.height + kMaxSb - 1) / kMaxSb; nsb = sb_cols * sb_rows;
    max_diag = 0;
    for (int t = 0; t <= (sb_cols - 1) + 2 * (sb_rows - 1); t++) {
      int n = 0;
      for (int k = 0; k < sb_rows; k++) { int l = t - 2 * k; if (l >= 0 && l < sb_cols) n++; }
      if (n > max_diag) max_diag = n;
    }
    nfb_h = (p.width + 63) >> 6; nfb_v = (p.height + 63) >> 6;
    ring_size = (p.HQperiod > p.max_num_ref ? p.HQperiod : p.max_num_ref) + 1;
    if (p.num_reorder_pics > 0) ring_size = 2 * (p.num_reorder_pics + 1) + 1 > ring_size ? 2 * (p.num_reorder_pics + 1) + 1 : ring_size;
    if (ring_size > 33) ring_size = 33;
    ws_bytes = (backend::team_ws_bytes((int)sizeof(PIX)) + 255) & ~(size_t)255;
    st.resize(S);
    d_nbits_all = (int*)backend::dev_alloc((size_t)S * nsb * sizeof(int));
    d_status_all = (int*)backend::dev_alloc((size_t)S * nsb * sizeof(int));
    d_items = (backend::GatherItem*)backend::dev_alloc((size_t)S * nsb * sizeof(backend::GatherItem));
    { int si = 0; for (auto& s : st) { s.sb_nbits = d_nbits_all + (size_t)si * nsb; s.sb_status = d_status_all + (size_t)si * nsb; si++; } }
    const int cw = p.width / 4, chh = p.height / 4;
    for (auto& s : st) {
      s.orig.alloc(p.width, p.height, 0);
      s.rec.alloc(p.width, p.height, 0);
      s.tmp.alloc(p.width, p.height, 0);
      s.ring.resize(ring_size);
      for (auto& r : s.ring) r.alloc(p.width, p.height, kPadY);
      if (p.interp_ref) s.interp.alloc(p.width, p.height, kPadY);
      if (p.interp_ref && !external_interp) {
        const int mn = p.width < p.height ? p.width : p.height;
        int lv = (int)(log10((double)mn) / log10(2.0) - 4.0);  // temporal_interp.c:913, evaluated in double as written
        if (lv > idev::kMaxLevels) lv = idev::kMaxLevels;
        if (lv < 1) { fprintf(stderr, "Run-time error...\nthor_hip: frame too small for interp_ref\n...now exiting to system...\n"); abort(); }
        s.ip_levels = lv;
        for (int l = 0; l < lv; l++) {
          const int lw = p.width >> l, lh = p.height >> l;
          const int bw = idev::kStep * ((lw + idev::kBbs - 1) / idev::kBbs), bh = idev::kStep * ((lh + idev::kBbs - 1) / idev::kBbs);
          const size_t n = (size_t)bw * bh + bw + 2;
          for (int k = 0; k < 5; k++) s.ip_mv[l][k] = (idev::imv*)backend::dev_alloc(n * sizeof(idev::imv));
          s.ip_prog[l] = (int*)backend::dev_alloc((size_t)(bh / idev::kStep + 1) * sizeof(int));
          if (l > 0) {
            s.ip_stride[l] = (lw + 64 + 15) & ~15;
            for (int k = 0; k < 2; k++) s.ip_pic[k][l] = (PIX*)backend::dev_alloc(((size_t)(lh + 64) * s.ip_stride[l] + 64) * sizeof(PIX));
          }
        }
      }
      s.cells = (DbCell*)backend::dev_alloc((size_t)cw * chh * sizeof(DbCell));
      backend::dev_memset(s.cells, 0, (size_t)cw * chh * sizeof(DbCell));
      s.sb_bits = (uint32_t*)backend::dev_alloc((size_t)nsb * kSbWords * 4);
      s.scratch = (uint8_t*)backend::dev_alloc(ws_bytes * max_diag);
      const int nfb = nfb_h * nfb_v;
      s.cdef_dir = (int8_t*)backend::dev_alloc((size_t)(p.width / 8) * (p.height / 8));
      s.cdef_var = (int*)backend::dev_alloc((size_t)(p.width / 8) * (p.height / 8) * sizeof(int));
      s.cdef_fbc = (int*)backend::dev_alloc(nfb * sizeof(int));
      s.cdef_mse = (unsigned long long*)backend::dev_alloc((size_t)2 * nfb * kCdefMaxStr * 8);
      s.cdef_sel = (int*)backend::dev_alloc(nfb * sizeof(int));
      s.cdef_fbsel = (int*)backend::dev_alloc(nfb * sizeof(int));
      s.cdef_res = (CdefResult*)backend::dev_alloc(sizeof(CdefResult));
      s.cdef_tot = (unsigned long long*)backend::dev_alloc(((size_t)kCdefMaxStr * kCdefMaxStr + nfb) * 8);
      if (p.clpf) {
        s.clpf_stats = (uint32_t*)backend::dev_alloc(clpf_stat_words() * 4);
        s.clpf_fb_on = (uint8_t*)backend::dev_alloc((size_t)((p.width + 31) / 32) * ((p.height + 31) / 32));
      }
      if (!raw_frames) write_sequence_header(s.bits, sp);
    }
    d_ljobs = (ClpfJob<PIX>*)backend::dev_alloc(sizeof(ClpfJob<PIX>) * S);
    h_ljobs.resize(S);
    d_jobs = (FrameJob<PIX>*)backend::dev_alloc(sizeof(FrameJob<PIX>) * S);
    h_jobs.resize(S);
    d_cjobs = (CdefJob<PIX>*)backend::dev_alloc(sizeof(CdefJob<PIX>) * S);
    h_cjobs.resize(S);
    d_prof = (long long*)backend::dev_alloc(32 * sizeof(long long));
    d_stats = (unsigned long long*)backend::dev_alloc(8 * sizeof(unsigned long long));
    backend::dev_memset(d_stats, 0, 8 * sizeof(unsigned long long));
    if (p.interp_ref && !external_interp) { d_ijobs = (idev::Job<PIX>*)backend::dev_alloc(sizeof(idev::Job<PIX>) * S); h_ijobs.resize(S); }
  }
  size_t clpf_stat_words() const { return 4 * ((size_t)(sp.width / 8) * (sp.height / 8) + 2 * (size_t)(sp.width / 16) * (sp.height / 16)); }
  void close() {
    for (auto& s : st) {
      backend::dev_free(s.clpf_stats); backend::dev_free(s.clpf_fb_on);
      s.orig.release(); s.rec.release(); s.tmp.release(); s.interp.release();
      backend::dev_free(s.cdef_dir); backend::dev_free(s.cdef_var); backend::dev_free(s.cdef_fbc); backend::dev_free(s.cdef_mse);
      backend::dev_free(s.cdef_sel); backend::dev_free(s.cdef_fbsel); backend::dev_free(s.cdef_res); backend::dev_free(s.cdef_tot);
      for (auto& r : s.ring) r.release();
      backend::dev_free(s.cells); backend::dev_free(s.sb_bits); backend::dev_free(s.scratch);
      for (int l = 0; l < idev::kMaxLevels; l++) {
        for (int k = 0; k < 5; k++) backend::dev_free(s.ip_mv[l][k]);
        backend::dev_free(s.ip_prog[l]); backend::dev_free(s.ip_pic[0][l]); backend::dev_free(s.ip_pic[1][l]);
      }
    }
    st.clear();
    backend::dev_free(d_nbits_all); backend::dev_free(d_status_all); backend::dev_free(d_items); backend::dev_free(d_payload);
    d_nbits_all = d_status_all = nullptr; d_items = nullptr; d_payload = nullptr; payload_words = 0;
    backend::release_superblocks(d_jobs);
    backend::dev_free(d_jobs); d_jobs = nullptr;
    backend::dev_free(d_cjobs); d_cjobs = nullptr;
    backend::dev_free(d_ljobs); d_ljobs = nullptr;
    backend::dev_free(d_prof); d_prof = nullptr;
    backend::dev_free(d_stats); d_stats = nullptr;
    backend::dev_free(d_ijobs); d_ijobs = nullptr;
  }

  // planar 4:2:0 frame in host memory -> device `orig` of stream s
  void upload_orig(int s, const PIX* yuv) {
    const int w = sp.width, h = sp.height;
    std::vector<PIX> tmp;
    DevFrame<PIX>& f = st[s].orig;
    if (f.p.sy == w) backend::h2d(f.p.y, yuv, (size_t)w * h * sizeof(PIX));
    else { tmp.assign((size_t)f.p.sy * h, 0); for (int i = 0; i < h; i++) memcpy(&tmp[(size_t)i * f.p.sy], yuv + (size_t)i * w, w * sizeof(PIX)); backend::h2d(f.p.y, tmp.data(), tmp.size() * sizeof(PIX)); }
    const PIX* cu = yuv + (size_t)w * h; const PIX* cv = cu + (size_t)(w / 2) * (h / 2);
    if (f.p.sc == w / 2) { backend::h2d(f.p.u, cu, (size_t)(w / 2) * (h / 2) * sizeof(PIX)); backend::h2d(f.p.v, cv, (size_t)(w / 2) * (h / 2) * sizeof(PIX)); }
    else {
      tmp.assign((size_t)f.p.sc * (h / 2), 0);
      for (int i = 0; i < h / 2; i++) memcpy(&tmp[(size_t)i * f.p.sc], cu + (size_t)i * (w / 2), (w / 2) * sizeof(PIX));
      backend::h2d(f.p.u, tmp.data(), tmp.size() * sizeof(PIX));
      for (int i = 0; i < h / 2; i++) memcpy(&tmp[(size_t)i * f.p.sc], cv + (size_t)i * (w / 2), (w / 2) * sizeof(PIX));
      backend::h2d(f.p.v, tmp.data(), tmp.size() * sizeof(PIX));
    }
  }
  // planes with arbitrary host strides (drop-in path)
  void upload_planes(int s, const PIX* y, int sy, const PIX* u, const PIX* v, int sc) {
    const int w = sp.width, h = sp.height;
    DevFrame<PIX>& f = st[s].orig;
    std::vector<PIX> tmp((size_t)f.p.sy * h);
    for (int i = 0; i < h; i++) memcpy(&tmp[(size_t)i * f.p.sy], y + (size_t)i * sy, w * sizeof(PIX));
    backend::h2d(f.p.y, tmp.data(), tmp.size() * sizeof(PIX));
    tmp.assign((size_t)f.p.sc * (h / 2), 0);
    for (int i = 0; i < h / 2; i++) memcpy(&tmp[(size_t)i * f.p.sc], u + (size_t)i * sc, (w / 2) * sizeof(PIX));
    backend::h2d(f.p.u, tmp.data(), tmp.size() * sizeof(PIX));
    for (int i = 0; i < h / 2; i++) memcpy(&tmp[(size_t)i * f.p.sc], v + (size_t)i * sc, (w / 2) * sizeof(PIX));
    backend::h2d(f.p.v, tmp.data(), tmp.size() * sizeof(PIX));
  }
  void download_rec(int s, PIX* yuv) {
    const int w = sp.width, h = sp.height;
    DevFrame<PIX>& f = st[s].rec;
    std::vector<PIX> tmp((size_t)f.p.sy * h);
    backend::d2h(tmp.data(), f.p.y, tmp.size() * sizeof(PIX));
    for (int i = 0; i < h; i++) memcpy(yuv + (size_t)i * w, &tmp[(size_t)i * f.p.sy], w * sizeof(PIX));
    PIX* cu = yuv + (size_t)w * h; PIX* cv = cu + (size_t)(w / 2) * (h / 2);
    tmp.resize((size_t)f.p.sc * (h / 2));
    backend::d2h(tmp.data(), f.p.u, tmp.size() * sizeof(PIX));
    for (int i = 0; i < h / 2; i++) memcpy(cu + (size_t)i * (w / 2), &tmp[(size_t)i * f.p.sc], (w / 2) * sizeof(PIX));
    backend::d2h(tmp.data(), f.p.v, tmp.size() * sizeof(PIX));
    for (int i = 0; i < h / 2; i++) memcpy(cv + (size_t)i * (w / 2), &tmp[(size_t)i * f.p.sc], (w / 2) * sizeof(PIX));
  }

  // The frame's per-4x4 block data (DbCell, tk_common.h) as the block decisions and the in-loop filters left it: what the
  // reference keeps in encoder_info->deblock_data[] (written by copy_deblock_data, enc/encode_block.c:1568-1613).
  size_t num_cells() const { return (size_t)(sp.width / 4) * (sp.height / 4); }
  void download_cells(int s, DbCell* out) { backend::d2h(out, st[s].cells, num_cells() * sizeof(DbCell)); }

  // Coding-order schedule.  begin_sequence fixes the chunk [skip, skip+num_frames) of an input holding
  // file_frames frames (needed for the reference's end-of-sequence behaviour with reordered GOPs).
  void begin_sequence(int s, int skip, int num_frames, int file_frames) { st[s].gop.init(sp, skip, num_frames, file_frames); }
  // Next frame to code for stream s: fills st[s].cur / st[s].cur_abs; false when the chunk is finished.
  bool schedule(int s) {
    Stream<PIX>& q = st[s];
    if (!q.gop.started) q.gop.init(sp, 0, 1 << 28, 1 << 28);
    return q.gop.next(q.cur, q.cur_abs, [&](int idx) { return q.ring[idx].frame_num; });
  }

  // Temporally interpolated reference (enc/mainenc.c:350-355, common/temporal_interp.c:909): built on the device from
  // the two window frames of every stream whose frame uses it (tk_interp_dev.h) - no frame leaves HBM.
  void make_interp_frames(const std::vector<FrameParams>& fp, int first, int count) {
    int n = 0;
    for (int s = first; s < first + count; s++) {
      if (!fp[s].interp_ref) continue;
      Stream<PIX>& q = st[s];
      const FrameParams& f = fp[s];
      idev::Job<PIX>& J = h_ijobs[n++];
      memset(&J, 0, sizeof(J));
      const int ratio = 2, k = 1;  // interpolate_frames(..., 2, 1): the frame half way between the two anchors (mainenc.c:353)
      J.levels = q.ip_levels;
      J.width = sp.width; J.height = sp.height;
      J.ref[0] = q.ring[f.interp_src[0]].p; J.ref[1] = q.ring[f.interp_src[1]].p; J.out = q.interp.p;
      const int reversed = k > ratio / 2;
      const int wt0 = reversed ? k : ratio - k, wt1 = ratio - wt0;
      for (int l = 0; l < J.levels; l++) {
        idev::Level<PIX>& L = J.lv[l];
        const int lw = sp.width >> l, lh = sp.height >> l;
        const PIX* in[2];
        int str[2];
        if (l == 0) { in[0] = J.ref[0].y; in[1] = J.ref[1].y; str[0] = J.ref[0].sy; str[1] = J.ref[1].sy; L.pad = kPadY; }
        else {
          for (int r = 0; r < 2; r++) { J.dpic[r][l] = q.ip_pic[r][l] + (size_t)32 * q.ip_stride[l] + 32; in[r] = J.dpic[r][l]; str[r] = q.ip_stride[l]; }
          J.dstride[l] = q.ip_stride[l];
          L.pad = 32;
        }
        L.pic[0] = reversed ? in[1] : in[0]; L.pic[1] = reversed ? in[0] : in[1];
        L.s[0] = reversed ? str[1] : str[0]; L.s[1] = reversed ? str[0] : str[1];
        L.w = lw; L.h = lh;
        L.mv[0] = q.ip_mv[l][0]; L.mv[1] = q.ip_mv[l][1]; L.nmv[0] = q.ip_mv[l][2]; L.nmv[1] = q.ip_mv[l][3]; L.gmv1 = q.ip_mv[l][4];
        L.guide_mv1 = l == J.levels - 1 ? nullptr : L.gmv1;
        L.guide_reversed = reversed; L.guide_wt0 = wt0;
        L.wt[0] = wt0; L.wt[1] = wt1; L.reversed = reversed;
        L.bw = idev::kStep * ((lw + idev::kBbs - 1) / idev::kBbs); L.bh = idev::kStep * ((lh + idev::kBbs - 1) / idev::kBbs);
        L.prog = q.ip_prog[l]; L.ticket = q.ip_prog[l] + L.bh / idev::kStep;
      }
      q.interp.frame_num = f.frame_num;
    }
    if (!n) return;
    backend::h2d(d_ijobs, h_ijobs.data(), sizeof(idev::Job<PIX>) * n);
    backend::run_interp<PIX>(d_ijobs, h_ijobs.data(), n);
  }

  // Encode one frame per stream (origs already uploaded). Appends to st[s].out.
  void encode_frames(const std::vector<FrameParams>& fp) {
    prepare_jobs(fp, 0, S);
    backend::run_superblocks<PIX>(d_jobs, h_jobs.data(), S, nullptr);
    finish_frames(fp, 0, S);
  }

  // Anti-diagonals of the superblock grid (t = l + 2k): their number, and where a frame is cut in two for encode_run.
  int num_diags() const { return (sb_cols - 1) + 2 * (sb_rows - 1) + 1; }

  // Code the next `nframes` frames of every stream, the streams in TWO GROUPS HALF A FRAME APART (the first half of the streams leads):
  // one launch of the persistent superblock kernel carries the second half (anti-diagonals [Th, T)) of one group's frame - the narrowing end
  // of its dependency wavefront - together with the first half of the other group's frame, its widening start, so the workgroup slots one
  // group leaves idle are taken by the other (in lock step every stream ramps up and down at the same time: 13 % of the workgroup-time idle
  // at 3840x2160 x 128 streams, profiles/r04_sbtimes_4k_s128_final.log).  The run starts and ends on frame boundaries for every stream: the
  // first launch holds only the leading group's first half, the last one only the trailing group's second half.  Every superblock still
  // starts after its two dependencies and sees the same reference frames: results do not depend on the schedule.
  //   next(s): make stream s ready for its next frame - schedule() it and point st[s].orig at its input; false: no frame left (an error here)
  //   done(first, count): the frames of streams [first, first + count) are complete (bits appended, st[s].rec = the reconstruction,
  //                       st[s].cur = the frame) - called before any later launch touches those streams again
  template <class NextF, class DoneF> void encode_run(int nframes, NextF next, DoneF done) {
    std::vector<FrameParams> fp(S);
    auto prep = [&](int first, int count) {
      for (int s = first; s < first + count; s++) {
        if (!next(s)) { fprintf(stderr, "thor_hip: stream %d has no frame left to code\n", s); abort(); }
        fp[s] = st[s].cur;
      }
      prepare_jobs(fp, first, count);
    };
    if (S < 2 || num_diags() < 2) {   // nothing to stagger: frame by frame
      for (int f = 0; f < nframes; f++) {
        prep(0, S);
        backend::run_superblocks<PIX>(d_jobs, h_jobs.data(), S, nullptr);
        finish_frames(fp, 0, S);
        done(0, S);
      }
      return;
    }
    const int T = num_diags(), Th = (T + 1) / 2, S0 = S / 2, S1 = S - S0;
    std::vector<backend::SbRange> rg(S);
    auto set = [&](int first, int count, int lo, int hi) { for (int s = first; s < first + count; s++) { rg[s].lo = (unsigned short)lo; rg[s].hi = (unsigned short)hi; } };
    if (nframes < 1) return;
    prep(0, S0);
    set(0, S0, 0, Th); set(S0, S1, 0, 0);
    backend::run_superblocks<PIX>(d_jobs, h_jobs.data(), S, rg.data());
    for (int f = 0; f < nframes; f++) {
      prep(S0, S1);
      set(0, S0, Th, T); set(S0, S1, 0, Th);
      backend::run_superblocks<PIX>(d_jobs, h_jobs.data(), S, rg.data());
      finish_frames(fp, 0, S0);
      done(0, S0);
      if (f + 1 < nframes) { prep(0, S0); set(0, S0, 0, Th); } else set(0, S0, 0, 0);
      set(S0, S1, Th, T);
      backend::run_superblocks<PIX>(d_jobs, h_jobs.data(), S, rg.data());
      finish_frames(fp, S0, S1);
      done(S0, S1);
    }
  }

  // Frame jobs of streams [first, first + count) for the frames fp[s] (origs in place): interpolated references, FrameJob set-up, upload.
  void prepare_jobs(const std::vector<FrameParams>& fp, int first, int count) {
    if (!external_interp) make_interp_frames(fp, first, count);
    for (int s = first; s < first + count; s++) {
      Stream<PIX>& q = st[s];
      const FrameParams& f = fp[s];
      FrameJob<PIX>& J = h_jobs[s];
      memset(&J, 0, sizeof(J));
      EncCfg& c = J.cfg;
      c.width = sp.width; c.height = sp.height; c.bitdepth = sp.bitdepth;
      c.enable_tb_split = sp.enable_tb_split; c.enable_pb_split = sp.enable_pb_split; c.enable_bipred = sp.enable_bipred;
      c.encoder_speed = sp.encoder_speed; c.intra_rdo = sp.intra_rdo; c.use_block_contexts = sp.use_block_contexts;
      c.cfl_intra = sp.cfl_intra; c.cfl_inter = sp.cfl_inter; c.max_num_ref = sp.max_num_ref; c.interp_ref_cfg = sp.interp_ref;
      c.early_skip_thr = sp.early_skip_thr;
      J.frame_type = f.frame_type; J.qp = f.qp; J.num_ref = f.num_ref; J.frame_num = f.frame_num;
      J.interp_ref = f.interp_ref; J.num_intra_modes = f.num_intra_modes;
      J.lambda = f.lambda_coeff * kSquaredLambdaQP[f.qp];
      J.sqrt_lambda = sqrt(J.lambda);
      J.orig = q.orig.p; J.rec = q.rec.p;
      for (int r = 0; r < f.num_ref; r++) {
        const DevFrame<PIX>& rf = f.ref_array[r] < 0 ? q.interp : q.ring[f.ref_array[r]];
        J.ref[r] = rf.p;
        J.sign[r] = rf.frame_num > f.frame_num;
        J.sign_ge[r] = rf.frame_num >= f.frame_num;
      }
      J.cells = q.cells; J.cell_stride = sp.width / 4;
      J.sb_cols = sb_cols; J.sb_rows = sb_rows;
      J.sb_bits = q.sb_bits; J.sb_words = kSbWords; J.sb_nbits = q.sb_nbits; J.sb_status = q.sb_status;
      J.scratch = q.scratch; J.scratch_bytes = ws_bytes;
      J.prof = d_prof;
      J.stats = d_stats;
      if (f.frame_type == F_I) backend::dev_memset(q.cells, 0, (size_t)(sp.width / 4) * (sp.height / 4) * sizeof(DbCell));
    }
    backend::h2d(d_jobs + first, h_jobs.data() + first, sizeof(FrameJob<PIX>) * count);
  }

  // Everything after the superblocks of streams [first, first + count): in-loop filters, reference creation, bitstream assembly.
  void finish_frames(const std::vector<FrameParams>& fp, int first, int count) {
    const int end = first + count;
    if (sp.deblocking) backend::run_deblock<PIX>(d_jobs + first, h_jobs.data() + first, count);
    // CDEF (encode_frame.c:685-689 frame-level guesses, :768-783 search + filter)
    if (sp.cdef) {
      for (int s = first; s < end; s++) {
        Stream<PIX>& q = st[s];
        CdefJob<PIX>& C = h_cjobs[s];
        C.rec = q.rec.p; C.src = q.tmp.p; C.org = q.orig.p;
        C.width = sp.width; C.height = sp.height; C.bitdepth = sp.bitdepth;
        C.cells = q.cells; C.cs = sp.width / 4; C.nfb_h = nfb_h; C.nfb_v = nfb_v;
        C.speed = sp.cdef - 1; C.damping = 5;
        C.cdef_bits = fp[s].frame_type == F_I ? 3 : 3 - (fp[s].qp + 4) / 16;
        if (C.speed == 3) C.cdef_bits = 0;
        C.qp = fp[s].qp;
        C.dir = q.cdef_dir; C.var = q.cdef_var; C.fb_compact = q.cdef_fbc; C.mse = q.cdef_mse; C.sel = q.cdef_sel;
        C.fb_sel = q.cdef_fbsel; C.res = q.cdef_res; C.tot = q.cdef_tot;
      }
      backend::h2d(d_cjobs + first, h_cjobs.data() + first, sizeof(CdefJob<PIX>) * count);
      backend::run_cdef<PIX>(d_cjobs + first, h_cjobs.data() + first, count);
    }
    // CLPF (encode_frame.c:785-817): statistics on the device, frame-level plan on the host, filter on the device
    std::vector<ClpfPlan> lplan(sp.clpf ? S : 0);
    if (sp.clpf) {
      bool any = false;
      for (int s = first; s < end; s++) {
        Stream<PIX>& q = st[s];
        ClpfJob<PIX>& L = h_ljobs[s];
        L.rec = q.rec.p; L.src = q.tmp.p; L.org = q.orig.p;
        L.width = sp.width; L.height = sp.height; L.bitdepth = sp.bitdepth; L.qp = fp[s].qp;
        L.cells = q.cells; L.cs = sp.width / 4; L.stats = q.clpf_stats;
        L.strength[0] = L.strength[1] = L.strength[2] = 0; L.fb_log2 = 7; L.fb_on = q.clpf_fb_on;
        any = any || fp[s].qp > 16;
      }
      if (any) {
        backend::h2d(d_ljobs + first, h_ljobs.data() + first, sizeof(ClpfJob<PIX>) * count);
        backend::run_clpf_stats<PIX>(d_ljobs + first, h_ljobs.data() + first, count);
        backend::dev_sync();
        std::vector<uint32_t> hst(clpf_stat_words());
        bool filt = false;
        for (int s = first; s < end; s++) {
          if (fp[s].qp > 16) backend::d2h(hst.data(), st[s].clpf_stats, hst.size() * 4);
          lplan[s] = clpf_plan(hst.data(), sp.width, sp.height, fp[s].qp, h_jobs[s].lambda, sp.max_clpf_strength);
          ClpfJob<PIX>& L = h_ljobs[s];
          for (int k = 0; k < 3; k++) { L.strength[k] = lplan[s].strength[k]; filt = filt || L.strength[k]; }
          L.fb_log2 = lplan[s].fb_log2;
          if (!lplan[s].fb_on.empty()) backend::h2d(st[s].clpf_fb_on, lplan[s].fb_on.data(), lplan[s].fb_on.size());
        }
        if (filt) {
          backend::h2d(d_ljobs + first, h_ljobs.data() + first, sizeof(ClpfJob<PIX>) * count);
          backend::run_clpf_apply<PIX>(d_ljobs + first, h_ljobs.data() + first, count);
        }
      } else
        for (int s = first; s < end; s++) lplan[s] = clpf_plan(nullptr, sp.width, sp.height, fp[s].qp, h_jobs[s].lambda, sp.max_clpf_strength);
    }
    // sliding window: the slot shifted out becomes ref[0] (encode_frame.c:826-835)
    std::vector<Plane3<PIX>> dst(S);
    for (int s = first; s < end; s++) {
      Stream<PIX>& q = st[s];
      DevFrame<PIX> last = q.ring.back();
      for (int r = ring_size - 1; r > 0; r--) q.ring[r] = q.ring[r - 1];
      q.ring[0] = last;
      q.ring[0].frame_num = fp[s].frame_num;
      dst[s] = q.ring[0].p;
    }
    backend::run_make_ref<PIX>(h_jobs.data() + first, dst.data() + first, count);
    backend::dev_sync();
    // bitstream assembly: one D2H of all bit counts, a device-side bit-level gather of the per-SB
    // strings into one compact buffer, one D2H of that buffer.
    // (arrays indexed by stream: only the entries of [first, end) are filled and used)
    std::vector<int> nb((size_t)S * nsb), stt((size_t)S * nsb);
    backend::d2h(nb.data() + (size_t)first * nsb, d_nbits_all + (size_t)first * nsb, (size_t)count * nsb * sizeof(int));
    backend::d2h(stt.data() + (size_t)first * nsb, d_status_all + (size_t)first * nsb, (size_t)count * nsb * sizeof(int));
    std::vector<backend::GatherItem> items((size_t)S * nsb);
    std::vector<long long> stream_off(S + 1, 0);
    {
      long long pos = 0;
      for (int s = first; s < end; s++) {
        stream_off[s] = pos;
        for (int i = 0; i < nsb; i++) {
          if (stt[(size_t)s * nsb + i]) { fprintf(stderr, "Run-time error...\nthor_hip: superblock %d bit buffer overflow\n...now exiting to system...\n", i); abort(); }
          backend::GatherItem& g = items[(size_t)s * nsb + i];
          g.src = st[s].sb_bits + (size_t)i * kSbWords; g.nbits = nb[(size_t)s * nsb + i]; g.dst_bit = pos;
          pos += g.nbits;
        }
        pos = (pos + 31) & ~31ll;  // streams start word aligned
      }
      stream_off[end] = pos;
    }
    const size_t need_words = (size_t)(stream_off[end] >> 5) + 2;
    if (need_words > payload_words) {
      backend::dev_free(d_payload);
      payload_words = need_words + need_words / 2;
      d_payload = (uint32_t*)backend::dev_alloc(payload_words * 4);
    }
    backend::dev_memset(d_payload, 0, need_words * 4);
    backend::h2d(d_items + (size_t)first * nsb, items.data() + (size_t)first * nsb, (size_t)count * nsb * sizeof(backend::GatherItem));
    backend::run_gather(d_items + (size_t)first * nsb, count * nsb, d_payload);
    std::vector<uint32_t> words(need_words);
    backend::d2h(words.data(), d_payload, need_words * 4);
    for (int s = first; s < end; s++) {
      Stream<PIX>& q = st[s];
      const FrameParams& f = fp[s];
      HostBits& b = q.bits;
      // write_frame_header (write_bits.c:98-121)
      b.put(1, f.frame_type != F_I); b.put(8, f.qp); b.put(4, f.num_intra_modes);
      if (f.frame_type != F_I) b.put(2, f.num_ref - 1);
      for (int r = 0; r < f.num_ref; r++) b.put(6, f.ref_array[r] + 1);
      b.put(16, f.frame_num);
      CdefHeader ch;
      const int cdef_pos = b.nbits;
      if (sp.cdef) {
        ch.bits = h_cjobs[s].cdef_bits;
        for (int i = 0; i < 8; i++) ch.strengths[i] = ch.uv_strengths[i] = 127;
      }
      write_cdef_params(b, -1, sp.cdef, ch);
      {
        long long pbits = 0;
        for (int i = 0; i < nsb; i++) pbits += nb[(size_t)s * nsb + i];
        b.append_words(words.data() + (stream_off[s] >> 5), (int)pbits);
      }
      if (sp.cdef) {
        CdefResult R;
        backend::d2h(&R, q.cdef_res, sizeof(R));
        if (h_cjobs[s].cdef_bits != 0 && R.nb_bits) {
          std::vector<int> sel(R.sb_count);
          if (R.sb_count) backend::d2h(sel.data(), q.cdef_sel, R.sb_count * sizeof(int));
          for (int i = 0; i < R.sb_count; i++) b.put(R.nb_bits, (uint32_t)sel[i]);
        }
        ch.bits = R.nb_bits;
        for (int i = 0; i < 8; i++) { ch.strengths[i] = R.strengths[i]; ch.uv_strengths[i] = R.uv_strengths[i]; }
        write_cdef_params(b, cdef_pos, 1, ch, raw_frames ? q.bit_phase : 0);
      }
      if (sp.clpf)
        for (auto& pr : lplan[s].bits) b.put(pr.first, pr.second);
      q.num_encoded++;
      if (q.gop.started) q.gop.advance(f);
      if (raw_frames) continue;
      // flush_all_bits framing (putbits.c:45-83)
      uint32_t nbytes = (uint32_t)b.num_bytes();
      for (int i = 0; i < 4; i++) q.out.push_back((uint8_t)(nbytes >> (24 - 8 * i)));
      b.to_bytes(q.out);
      b.clear();
    }
  }
};

}  // namespace tk
