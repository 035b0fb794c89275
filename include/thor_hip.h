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
 * callback returns.  Returns 0; 2 when a stream has fewer than `nframes` frames left in its chunk, 3 when a frame of the run is not staged -
 * both are found by a dry run of every stream's schedule BEFORE the first launch: a refused run touches nothing and may be repeated. */
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

/* ---- (3b) round 6: known-answer entry points for the remaining sample kernels.  `bitdepth` 8: samples are uint8_t (the reference's _lbd
 * instances); 9..12: uint16_t (_hbd).  Every call runs the device function the encoder itself calls, one wavefront per item; known answers are
 * recorded from the reference functions named below (tests/golden/gen_kat5.py -> kat5.npz).  Return 0, 1 = bad argument, 2 = an item reads outside
 * the given frame, 3 = no usable device. */
/* make_top_and_left + get_intra_prediction (common/intra_prediction.c:57-183, :403-428) of `n` transform units of `size` x `size` samples.
 * plane: a reconstructed plane (stride in samples).  par[7*i..]: ypos, xpos of the CODING block, its upright / downleft availability, the intra
 * mode (0..9), and the unit's offset (i, j) inside the block.  tb_split 0: the unit IS the block (i = j = 0).  tb_split 1: the block is
 * 2*size wide, rblocks holds `n` block-local reconstructions of (2*size)^2 samples (what the units coded earlier left there).  out: n*size*size. */
int thor_hip_kat_intra(const void* plane, int width, int height, int stride, int bitdepth, int size, int tb_split, int n, const int* par,
                       const void* rblocks, void* out);
/* get_inter_prediction_yuv (common/inter_prediction.c:185-226: clip_mv, quarter-pel luma, eighth-pel chroma, one PU or four quadrant PUs) of `n`
 * blocks of `size` from a planar 4:2:0 reference frame (the library pads it like a reference picture).  par[5*i..]: ypos, xpos, sign,
 * enable_bipred, split; mv[8*i..]: four vectors (x, y).  out: n blocks of size*size luma + 2 * (size/2)^2 chroma samples. */
int thor_hip_kat_inter_yuv(const void* yuv, int width, int height, int bitdepth, int size, int n, const int* par, const int16_t* mv, void* out);
/* average_blocks_all (common/inter_prediction.c:228-247): truncating average of two yuv blocks (layout as above). */
int thor_hip_kat_average(const void* a, const void* b, int size, int bitdepth, int n, void* out);
/* improve_uv_prediction (common/common_block.c:347-428), 4:2:0: y = luma prediction, ry = reconstructed luma (both n_luma^2, stride n_luma),
 * uv = chroma predictions U then V ((n_luma/2)^2 each), updated in place. */
int thor_hip_kat_cfl(const void* y, void* uv, const void* ry, int n_luma, int bitdepth, int n);
/* cdef_find_dir (common/common_block.c:94-162) on `n` 8x8 blocks (64 samples each, coeff_shift = bitdepth - 8). */
int thor_hip_kat_cdef_dir(const void* blocks, int bitdepth, int n, int* dir, int* var);
/* cdef_filter_block (common/common_block.c:224-279) of `n` bsize x bsize blocks (8 luma, 4 chroma) of a plane; samples outside the plane count as
 * CDEF_VERY_LARGE (cdef_prepare_input, common/common_frame.c:766).  par[7*i..]: x0, y0, pri_strength, sec_strength, dir, pri_damping, sec_damping. */
int thor_hip_kat_cdef_filter(const void* plane, int width, int height, int stride, int bitdepth, int bsize, int n, const int* par, void* out);
/* CLPF on one planar 4:2:0 frame: the per-8x8-block squared errors of detect_multi_clpf (enc/encode_block.c:2556-2621; stats[4*b..]: unfiltered,
 * strength 1, 2, 4; blocks in luma raster order, then U, then V; a block whose top-left cell is skip-coded reports {0, 0xffffffff, 0, 0}) and the
 * filtered frame of clpf_frame / clpf_block (common/common_frame.c:1005-1163, common/common_block.c:325-345) for the given per-plane strengths
 * (0 = off, else 1 / 2 / 4), luma filter-block size 1 << fb_log2 and per-filter-block switches fb_on. */
int thor_hip_kat_clpf(const void* rec_yuv, const void* org_yuv, int width, int height, int bitdepth, int qp, const thor_hip_cell* cells,
                      const int* strength, int fb_log2, const uint8_t* fb_on, uint32_t* stats, void* out_yuv);
/* interpolate_frames(new, ref0, ref1, 2, 1) (common/temporal_interp.c:909: the frame half way between two reference pictures; luma pyramid,
 * block motion estimation per level, merge, motion-compensated average) through the engine's own device path (tk_interp_dev.h). */
int thor_hip_kat_interpolate(const void* yuv0, const void* yuv1, int width, int height, int bitdepth, void* out_yuv);

/* Resources of the persistent superblock kernel (sample_bytes 1: 8-bit kernel, 2: 16-bit kernel, 0: the 8-bit kernel's second build for runs of few
 * streams - 256 registers, two workgroups per CU, chosen by the library when a run cannot fill more; THOR_HIP_KERNEL=std|lat|wide forces one; 3: its third build - eight wavefronts per workgroup, one workgroup per CU, for runs of very few streams) as the HIP runtime reports them: registers per lane,
 * static LDS and private (scratch) bytes, and how many workgroups of it fit one CU - what the resident-workgroup count of every launch derives from. */
int thor_hip_superblock_kernel_info(int sample_bytes, int* num_regs, int* lds_bytes, int* private_bytes, int* workgroups_per_cu);
/* The build of the 8-bit superblock kernel the most recently configured engine launches: 0 = throughput build (168 registers, three four-wavefront
 * workgroups per CU), 1 = latency build (sample_bytes 0 above), 2 = eight wavefronts per workgroup, one workgroup per CU (sample_bytes 3 above: chosen while
 * all streams of a run together offer at most 2.5 superblocks per CU; the latency build runs only when THOR_HIP_KERNEL=lat forces it).  Valid after the engine's first encode call. */
int thor_hip_superblock_kernel_in_use(void);

#ifdef __cplusplus
}
#endif
#endif
