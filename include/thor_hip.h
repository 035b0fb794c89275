/* thor_hip.h - C ABI of libthor_hip.so, the MI355X (gfx950) implementation of the Thor encoder's
 * per-block hot path.  Plain C: pointers and sizes only, no C++/torch types.
 *
 * Three groups of entry points:
 *
 *  (1) The drop-in seam.  encode_frame_lbd / encode_frame_hbd are the only symbols the reference
 *      front end (enc/mainenc.o) imports from the hot path (enc/encode_frame.h:32-33, called at
 *      enc/mainenc.c:548).  Linking the reference's mainenc/strings/putbits/putvlc/write_bits
 *      objects against this library instead of encode_frame.o/encode_block.o gives a Thorenc that
 *      runs the block path on the GPU.  They take the reference's own
 *      encoder_info_t (enc/mainenc.h:158-184); its layout is restated in thor_abi.h.
 *
 *  (2) The sequence API used by bench.py / the tests: N independent closed streams (what the
 *      reference produces with -skip c*F -n F) encoded in lock step on one GPU,
 *      with the low-delay GOP decisions of enc/mainenc.c:261-523 made on the host.
 *
 *  (3) Kernel-level batch entry points for known-answer tests against the scalar reference
 *      functions (sad_calc encode_block.c:417, get_inter_prediction_luma inter_prediction.c:117,
 *      transform/quantize/dequantize/inverse_transform, deblock_frame_y/uv common_frame.c:47/354).
 *
 * Error convention: like the reference's fatalerror() (common/global.h:38-44) unrecoverable
 * conditions (no GPU, HIP failure, bit-buffer overflow) print to stderr and abort(); the
 * sequence API additionally returns 0 on success / non-zero on bad arguments.
 */
#ifndef THOR_HIP_H
#define THOR_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- (1) drop-in seam ------------------------------------------------------------------- */
struct thor_encoder_info; /* == reference encoder_info_t, see thor_abi.h */
void encode_frame_lbd(struct thor_encoder_info* encoder_info); /* replaces enc/encode_frame.c:637 (SAMPLE=uint8_t)  */
void encode_frame_hbd(struct thor_encoder_info* encoder_info); /* replaces enc/encode_frame_hbd.c (SAMPLE=uint16_t) */

/* ---- (2) sequence API --------------------------------------------------------------------- */
typedef struct thor_hip_params { /* the enc_params fields this path honours (enc/mainenc.h:35-112) */
  int width, height, qp;
  int bitdepth, input_bitdepth;
  float frame_rate;
  float lambda_coeffI, lambda_coeffP;
  float early_skip_thr;
  int enable_tb_split, enable_pb_split, max_num_ref, HQperiod;
  int num_reorder_pics, interp_ref;
  int dqpP, dqpI;
  float mqpP;
  int intra_period, intra_rdo, encoder_speed;
  int deblocking, cdef, clpf, use_block_contexts, enable_bipred;
  int cfl_intra, cfl_inter;
  /* hierarchical-B coding (num_reorder_pics > 0; enc/mainenc.c:270-345) */
  int dyadic_coding;
  float lambda_coeffB, lambda_coeffB0, lambda_coeffB1, lambda_coeffB2, lambda_coeffB3;
  int dqpB, dqpB0, dqpB1, dqpB2, dqpB3;
  float mqpB, mqpB0, mqpB1, mqpB2, mqpB3;
  int max_clpf_strength; /* CLPF strength cap (enc/strings.c:351) */
} thor_hip_params;

typedef struct thor_hip_encoder thor_hip_encoder;

/* Fill *p with the reference defaults (enc/strings.c:287-356) and then apply a Thorenc config
 * file ("-name value" tokens, ';' comments).  cfg_path may be NULL. Returns 0 on success. */
int thor_hip_params_from_config(thor_hip_params* p, const char* cfg_path);

/* Apply one "-name value" option (same names as enc/strings.c:287-356) on top of *p.  Returns 0 when the option is
 * honoured; 1 = not an option of the reference's table; 2 = known, but the value is not implemented by this path
 * (quantisation matrices, delta-QP / rate control, sync, subsampling other than 4:2:0, SB size other than 128: they
 * change the bitstream, so they are rejected instead of ignored); 3 = front-end option (-if/-of/-rf/-n/-skip).
 * thor_hip_params_from_config applies the same rule to every token of the file (returns non-zero). */
int thor_hip_params_set(thor_hip_params* p, const char* name, const char* value);

int thor_hip_device_count(void);
/* One process drives ONE GPU from ONE thread (the reference's encode_frame is neither re-entrant nor threaded,
 * SURVEY.md 8b): the first thor_hip_open binds the process to `device`; NULL is returned - with a message on stderr -
 * for a device index the node does not have, for a second device in the same process, and for parameter sets this
 * path cannot encode bit-exactly.  Multi-GPU = one process per GPU (bench.py under torch.distributed.run). */
thor_hip_encoder* thor_hip_open(const thor_hip_params* p, int num_streams, int device);
void thor_hip_close(thor_hip_encoder* e);

/* Coding-order schedule (the frame loop of enc/mainenc.c:246-625).  thor_hip_begin_sequence fixes the
 * chunk [skip, skip + num_frames) of an input holding file_frames frames for one stream; without it a
 * stream is an open-ended low-delay sequence starting at frame 0.  thor_hip_next_frame returns 1 and the
 * chunk-relative DISPLAY index of the next frame to code (B frames are coded out of display order), or 0
 * when the chunk is finished; the following encode call must be given exactly that frame.  Calling it is
 * optional for low-delay streams (coding order == display order). */
int thor_hip_begin_sequence(thor_hip_encoder* e, int stream, int skip, int num_frames, int file_frames);
int thor_hip_next_frame(thor_hip_encoder* e, int stream, int* display_index);

/* Copy one planar 4:2:0 frame (bitdepth 8: bytes; >8: little-endian uint16) of stream `stream`
 * into HBM staging slot `slot` (slots are allocated on demand). */
int thor_hip_stage_frame(thor_hip_encoder* e, int stream, int slot, const void* yuv);
/* Same, for a frame that is already in HBM (`dev_yuv` is a device pointer to the same contiguous planar layout):
 * device-to-device copy on the library's stream, synchronous at return.  The caller must have finished writing the
 * source (e.g. torch.cuda.synchronize()).  Used by bench.py: rank 0 broadcasts the clip over RCCL and every rank cuts
 * its chunks' frames out of it without a host round trip. */
int thor_hip_stage_frame_device(thor_hip_encoder* e, int stream, int slot, const void* dev_yuv);
/* Encode the next frame of every stream from staging slot slots[stream] (inputs already resident
 * in HBM).  Blocks until the bits of all streams are assembled on the host. */
int thor_hip_encode_staged(thor_hip_encoder* e, const int* slots);
/* Encode the next `nframes` frames of every stream, each from the staging slot with the frame's chunk-relative DISPLAY index (what
 * thor_hip_next_frame reports; low-delay streams: the coded index), with the streams in TWO GROUPS HALF A FRAME APART: a launch of the
 * persistent superblock kernel carries the second half of one group's frame (the narrowing end of its dependency wavefront over the
 * superblock grid) and the first half of the other group's, so the workgroup slots one group leaves idle are taken by the other - in lock
 * step (thor_hip_encode_staged frame by frame) every stream ramps up and down at the same time.  The run starts and ends on a frame
 * boundary of every stream; the bits and reconstructions are those of the frame-by-frame calls (the order in which superblocks are coded
 * never changes a result: every superblock starts after its dependencies, enc/encode_frame.c:697-835 raster order).
 * `done` (may be NULL) is called from the calling thread whenever the frames of streams [first_stream, first_stream + num_streams) are
 * complete: their bits are appended, thor_hip_get_recon returns that frame and thor_hip_last_display_index its display index until the
 * callback returns.  Returns 0, or non-zero when a stream has no frame left / a slot is not staged. */
typedef void (*thor_hip_frames_done_fn)(void* user, int first_stream, int num_streams);
int thor_hip_encode_staged_run(thor_hip_encoder* e, int nframes, thor_hip_frames_done_fn done, void* user);
/* Chunk-relative display index of the frame stream `stream` coded last (-1: none yet). */
int thor_hip_last_display_index(const thor_hip_encoder* e, int stream);
/* Convenience: stage + encode one host frame per stream (PCIe inclusive). */
int thor_hip_encode_frame(thor_hip_encoder* e, const void* const* yuv_per_stream);

/* Finished bitstream of a stream so far (sequence header + framed frames, identical to the file
 * the reference writes with -of). The pointer stays valid until the next encode call. */
size_t thor_hip_stream_bytes(const thor_hip_encoder* e, int stream);
const uint8_t* thor_hip_stream_data(const thor_hip_encoder* e, int stream);
/* Reconstruction of the most recently CODED frame of a stream (the reference writes these with -rf in
 * display order). */
int thor_hip_get_recon(thor_hip_encoder* e, int stream, void* yuv_out);

/* HIP-event time (ms) and launch count of the superblock kernel accumulated since the last
 * reset; used by bench.py for the roofline figure. */
void thor_hip_kernel_time(thor_hip_encoder* e, double* sb_ms, long* sb_launches, double* filter_ms);
void thor_hip_kernel_time_reset(thor_hip_encoder* e);
/* Content statistics of the inter frames coded since the last reset ("always log the fraction of SBs that
 * early-skipped"; reference shortcut enc/encode_block.c:2231,2440-2480): out[0] luma pixels coded by the early-skip
 * shortcut (any block size), out[1] superblocks early-skipped as a whole, out[2] superblocks processed, out[3] luma
 * pixels processed.  reset != 0 clears the counters. */
void thor_hip_read_stats(thor_hip_encoder* e, unsigned long long out[4], int reset);
/* Development aid: 32 shader-cycle / event counters summed over all superblock wavefronts (all zero unless
 * the library was built with -DTHOR_PROF). */
void thor_hip_read_prof(thor_hip_encoder* e, long long out[32]);

/* ---- (3) kernel-level batch entry points (known-answer tests) ------------------------------- */
/* SAD of `n` candidate positions: org is a compact w x h block (stride w); ref points at the
 * frame sample co-located with the block, stride rstride; cand[2*i],cand[2*i+1] = full-pel (dx,dy).
 * w, h: powers of two >= 4 (PU sizes).  Follows sad_calc (enc/encode_block.c:417-428). out[i] = SAD. */
int thor_hip_sad_batch(const uint8_t* org, int w, int h, const uint8_t* ref_plane, int plane_w, int plane_h,
                       int rstride, int bx, int by, const int* cand, int n, uint32_t* out);
/* Quarter-pel luma prediction (get_inter_prediction_luma, common/inter_prediction.c:117-181) of a
 * w x h block at (bx,by) for `n` motion vectors (mv[2*i]=x, mv[2*i+1]=y in 1/4 pel); out: n*w*h. */
int thor_hip_interp_luma(const uint8_t* ref_plane, int plane_w, int plane_h, int rstride, int pad, int bx, int by,
                         int w, int h, const int16_t* mv, int n, int bipred, uint8_t* out);
/* residual -> transform -> quantize -> dequantize -> inverse -> reconstruct of `n` size x size
 * blocks (transform.c:245, encode_block.c:84, common_block.c:45/75, transform.c:467).
 * org/pred/rec: n*size*size samples; coefq: n*min(size,16)^2; cbp: n flags. */
int thor_hip_code_tu_batch(const uint8_t* org, const uint8_t* pred, int size, int qp, int coeff_type, int fast, int n,
                           int16_t* coefq, uint8_t* rec, int* cbp);
/* In-loop deblocking of one 8-bit planar 4:2:0 frame in place (deblock_frame_y + deblock_frame_uv,
 * common/common_frame.c:47,354; chroma QP through chroma_qp[]).  cells: (height/4) x (width/4) records, the
 * device-side compact form of deblock_data_t (common/types.h:178-187): mode = block_mode_t, size = CB size,
 * tbpb = tb_split | pb_part << 1, cbp bit 0/1/2 = Y/U/V coefficients present. */
typedef struct thor_hip_cell {
  int16_t mv0x, mv0y, mv1x, mv1y;
  uint8_t mode, size, tbpb, cbp;
  int8_t ref0, ref1, dir;
  uint8_t pad;
} thor_hip_cell;
int thor_hip_deblock_frame(uint8_t* yuv, int width, int height, int qp, const thor_hip_cell* cells);

/* The same four entry points on 16-bit samples = the reference's _hbd instances (SAMPLE = uint16_t: enc/enc_kernels_hbd.c,
 * common/inter_prediction_hbd.c, common/common_block_hbd.c, common/common_frame_hbd.c); bitdepth 9..12, strides and sizes in
 * samples.  Known answers recorded from those reference functions: tests/golden/kat4.npz (bitdepth 10). */
int thor_hip_sad_batch_hbd(const uint16_t* org, int w, int h, const uint16_t* ref_plane, int plane_w, int plane_h,
                           int rstride, int bx, int by, const int* cand, int n, uint32_t* out);
int thor_hip_interp_luma_hbd(const uint16_t* ref_plane, int plane_w, int plane_h, int rstride, int pad, int bx, int by,
                             int w, int h, const int16_t* mv, int n, int bipred, int bitdepth, uint16_t* out);
int thor_hip_code_tu_batch_hbd(const uint16_t* org, const uint16_t* pred, int size, int qp, int coeff_type, int fast, int n,
                               int bitdepth, int16_t* coefq, uint16_t* rec, int* cbp);
int thor_hip_deblock_frame_hbd(uint16_t* yuv, int width, int height, int qp, int bitdepth, const thor_hip_cell* cells);

#ifdef __cplusplus
}
#endif
#endif
