/* refshim.c - TEST INFRASTRUCTURE.  Compiled only by oracle/Makefile (target reflib) when the
 * reference tree is present.  It #includes the reference's enc/encode_block.c BY PATH (nothing is
 * copied into this repository) to reach its file-static kernels, and exports thin wrappers so that
 * tests/golden/gen_kat.py can record known-answer vectors from the real reference code. */
#define STR2(x) #x
#define STR(x) STR2(x)
#include STR(REFDIR/enc/encode_block.c)
#include "simd.h"

void ref_init(int simd) { use_simd = simd; }
unsigned ref_sad_calc(uint8_t* a, uint8_t* b, int astride, int bstride, int w, int h) { return sad_calc(a, b, astride, bstride, w, h); }
int ref_quantize(int16_t* coeff, int16_t* coeffq, int qp, int size, int coeff_block_type) {
  return quantize(coeff, coeffq, qp, size, coeff_block_type, NULL);
}
int ref_quote_mv_bits(int dy, int dx) { return quote_mv_bits(dy, dx); }
/* widesad_calc (enc/encode_block.c:430-453, file-static): SAD at the five horizontal offsets -3 -1 0 1 3, the best one and its offset; 16x16 with use_simd
 * takes widesad_calc_simd (enc/enc_kernels.c:84-113), which needs `a` 16-byte aligned */
unsigned ref_widesad_calc(uint8_t* a, uint8_t* b, int astride, int bstride, int w, int h, int* x) { return widesad_calc(a, b, astride, bstride, w, h, x); }
/* bit length of write_coeff (enc/write_bits.c:145) for one coefficient block */
int ref_coeff_bits(int16_t* coeff, int size, int type) {
  static uint8_t buf[1 << 16];
  stream_t s;
  s.bytesize = sizeof buf; s.bytepos = 0; s.bitstream = buf; s.bitbuf = 0; s.bitrest = 32;
  write_coeff(&s, coeff, size, type);
  return get_bit_pos(&s);
}

/* The two sub-block tests of the early-skip check (enc/encode_block.c:2146-2229, file-static): luma = 2x2 average + (N/2)-point transform against half the
 * threshold, chroma = calc_cbp on the residual's column sums - with use_simd (what the encoder runs) calc_cbp_simd (enc/enc_kernels.c:828-907).  They read
 * params->bitdepth only.  pblock is a compact size x size block. */
int ref_early_skip_sub(int chroma, uint8_t* orig, int ostride, int size, int qp, uint8_t* pblock, float thr, int bitdepth, int simd) {
  static enc_params p;
  static encoder_info_t ei;
  memset(&p, 0, sizeof p);
  memset(&ei, 0, sizeof ei);
  p.bitdepth = bitdepth;
  ei.params = &p;
  use_simd = simd;
  return chroma ? check_early_skip_sub_blockC(&ei, orig, ostride, size, qp, pblock, thr) : check_early_skip_sub_block(&ei, orig, ostride, size, qp, pblock, thr);
}

/* encoder_speed > 0 sub-pel approximations (file-static in encode_block.c) and their SIMD twins */
unsigned ref_fasthalf(uint8_t* a, uint8_t* b, int as, int bs, int w, int h, int* x, int* y, int simd) {
  if (simd) return sad_calc_fasthalf_simd_lbd(a, b, as, bs, w, h, x, y);
  return sad_calc_fasthalf(a, b, as, bs, w, h, x, y);
}
unsigned ref_fastquarter(uint8_t* o, uint8_t* r, int os, int rs, int w, int h, int* x, int* y, int simd) {
  if (simd) return sad_calc_fastquarter_simd_lbd(o, r, os, rs, w, h, x, y);
  return sad_calc_fastquarter(o, r, os, rs, w, h, x, y);
}
int ref_clpf_sample(int X, int A, int B, int C, int D, int E, int F, int G, int H, int s, unsigned dmp) { return clpf_sample(X, A, B, C, D, E, F, G, H, s, dmp); }
void ref_detect_multi_clpf(const uint8_t* rec, const uint8_t* org, int x0, int y0, int width, int height, int ostride, int rstride, int* sum,
                           unsigned shift, unsigned size, unsigned dmp, int simd) {
  if (simd) detect_multi_clpf_simd_lbd(rec, org, x0, y0, width, height, ostride, rstride, sum, shift, size, dmp);
  else detect_multi_clpf_lbd(rec, org, x0, y0, width, height, ostride, rstride, sum, shift, size, dmp);
}
