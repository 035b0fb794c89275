/* thor_abi.h - binary layout of the reference structures that cross the drop-in seam
 * encode_frame_lbd / encode_frame_hbd (include/thor_hip.h group 1).
 *
 * These are layout-compatible restatements (x86-64 SysV) of the caller-owned structures the
 * reference front end (enc/mainenc.c) hands to encode_frame:
 *     yuv_frame_t      common/types.h:58-80      ->  thor_yuv_frame      (96 bytes)
 *     stream_t         enc/putbits.h:32-39       ->  thor_stream         (32 bytes)
 *     enc_params       enc/mainenc.h:35-112      ->  thor_enc_params
 *     frame_info_t     enc/mainenc.h:140-156     ->  thor_frame_info     (9040 bytes)
 *     encoder_info_t   enc/mainenc.h:158-184     ->  thor_encoder_info   (16456 bytes)
 *     deblock_data_t   common/types.h:178-187    ->  thor_deblock_data   (364 bytes; mv_t :132-136, inter_pred_t :138-145, cbp_t :147-152)
 * Only the members this path reads or writes are named individually; the rest are kept as opaque
 * storage of the right size.  tests/test_abi.py compiles a probe against the reference headers
 * (when /root/reference is present) and checks every named offset and every sizeof.
 */
#ifndef THOR_ABI_H
#define THOR_ABI_H
#include <stdint.h>

#define THOR_MAX_REF_FRAMES 33
#define THOR_MAX_SKIP_FRAMES 8

typedef struct thor_yuv_frame {
  void* y;
  void* u;
  void* v; /* sample (0,0) of each plane; uint8_t or uint16_t samples */
  int width, height;
  int stride_y, stride_c;
  int offset_y, offset_c;
  int pad_hor_y, pad_hor_c, pad_ver_y, pad_ver_c;
  int area_y, area_c;
  int sub, subsample;
  int frame_num;
  int bitdepth, input_bitdepth;
} thor_yuv_frame;

typedef struct thor_stream {
  uint32_t bytesize;
  uint32_t bytepos;
  uint8_t* bitstream;
  uint32_t bitbuf;
  uint32_t bitrest;
} thor_stream;

typedef struct thor_stream_pos {
  uint32_t bytepos, bitbuf, bitrest;
} thor_stream_pos;

typedef struct thor_enc_params {
  unsigned int width, height;
  int log2_sb_size;
  unsigned int qp;
  char *infilestr, *outfilestr, *reconfilestr, *statfilestr;
  unsigned int file_headerlen, frame_headerlen;
  int num_frames, skip;
  float frame_rate;
  float lambda_coeffI, lambda_coeffP, lambda_coeffB, lambda_coeffB0, lambda_coeffB1, lambda_coeffB2, lambda_coeffB3;
  float early_skip_thr;
  int enable_tb_split, enable_pb_split, max_num_ref, HQperiod, num_reorder_pics, dyadic_coding, interp_ref;
  int dqpP, dqpB, dqpB0, dqpB1, dqpB2, dqpB3;
  float mqpP, mqpB, mqpB0, mqpB1, mqpB2, mqpB3;
  int dqpI, intra_period, intra_rdo, max_delta_qp, delta_qp_step, encoder_speed, sync, deblocking;
  int cdef;
  int clpf, snrcalc, use_block_contexts, enable_bipred, bitrate, max_qp, min_qp, max_qpI, min_qpI;
  int qmtx, qmtx_offset, subsample, aspectnum, aspectden, max_clpf_strength, cfl_intra, cfl_inter;
  int bitdepth, frame_bitdepth, input_bitdepth;
} thor_enc_params;

typedef struct thor_mv {
  int16_t x, y;
} thor_mv;

typedef struct thor_frame_info {
  int frame_type; /* 0 I, 1 P, 2 B */
  uint8_t qp;
  int num_ref;
  int best_ref;
  int ref_array[THOR_MAX_REF_FRAMES];
  thor_mv mvcand[THOR_MAX_REF_FRAMES][64];
  int mvcand_num[THOR_MAX_REF_FRAMES];
  uint64_t mvcand_mask[THOR_MAX_REF_FRAMES];
  double lambda;
  int num_intra_modes;
  int frame_num;
  int interp_ref;
  int b_level;
  double lambda_coeff;
  int prev_qp;
  int min_ref_dist;
  int phase;
  int max_clpf_strength;
} thor_frame_info;

/* one entry per 4x4 block of the frame, row-major, (height/4) x (width/4) entries (enc/mainenc.c:208) */
typedef struct thor_inter_pred {
  thor_mv mv0, mv1;
  uint32_t ref_idx0, ref_idx1, bipred_flag;
} thor_inter_pred;
typedef struct thor_deblock_data {
  int mode;                /* block_mode_t: 0 skip, 1 intra, 2 inter, 3 bipred, 4 merge */
  int cbp_y, cbp_u, cbp_v; /* cbp_t */
  uint8_t size;
  uint8_t tb_split;
  int pb_part;             /* part_t */
  thor_inter_pred inter_pred;
  thor_inter_pred inter_pred_arr[16]; /* interp_ref 2 only (not supported by this path): never touched */
} thor_deblock_data;

typedef struct thor_encoder_info {
  void* block_info;
  thor_frame_info frame_info;
  thor_enc_params* params;
  thor_yuv_frame* orig;
  thor_yuv_frame* rec;
  thor_yuv_frame* tmp;
  thor_yuv_frame* ref[THOR_MAX_REF_FRAMES];
  thor_yuv_frame* interp_frames[THOR_MAX_SKIP_FRAMES];
  thor_stream* stream;
  thor_deblock_data* deblock_data;
  void* rc;
  int width, height, depth;
  void* wmatrix[12][3][2][6];
  void* iwmatrix[12][3][2][6];
  void* cdef;
  int cdef_damping;
  int cdef_bits;
  int cdef_strengths[8];
  int cdef_uv_strengths[8];
  thor_stream_pos cdef_header_pos;
} thor_encoder_info;

#if defined(__x86_64__) && defined(__GNUC__)
_Static_assert(sizeof(thor_yuv_frame) == 96, "yuv_frame_t layout");
_Static_assert(sizeof(thor_frame_info) == 9040, "frame_info_t layout");
_Static_assert(sizeof(thor_encoder_info) == 16456, "encoder_info_t layout");
_Static_assert(sizeof(thor_deblock_data) == 364, "deblock_data_t layout");
#endif
#endif
