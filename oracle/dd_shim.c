/* dd_shim.c - TEST INFRASTRUCTURE (oracle/): records the deblock_data[] array an encode_frame call leaves behind.
 *
 * Defines encode_frame_lbd / encode_frame_hbd (enc/encode_frame.h:32-33), forwards to the implementation under test and appends
 * the caller-visible fields of encoder_info->deblock_data[] (common/types.h:178-187, written by copy_deblock_data,
 * enc/encode_block.c:1568-1613) to the file named by $THOR_DD_DUMP: per frame a header {0x44444444, frame_num, cells} and 14
 * int32 per 4x4 cell {mode, cbp.y, cbp.u, cbp.v, size, tb_split, pb_part, mv0.x, mv0.y, mv1.x, mv1.y, ref_idx0, ref_idx1, bipred_flag}.
 *   -DDD_STATIC : the implementation is the reference's own object with its symbols renamed by objcopy (Thorenc_dd)
 *   otherwise   : the next definition in link order, i.e. libthor_hip.so (Thorenc_hip_dd)
 * Compiled against the reference headers where they lie; never part of the product. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "global.h"
#include "mainenc.h"

#ifdef DD_STATIC
void ref_encode_frame_lbd(encoder_info_t *);
void ref_encode_frame_hbd(encoder_info_t *);
#endif

static void dd_dump(const encoder_info_t *e) {
  const char *path = getenv("THOR_DD_DUMP");
  if (!path) return;
  FILE *f = fopen(path, "ab");
  if (!f) { perror(path); abort(); }
  const int cells = (e->height / MIN_PB_SIZE) * (e->width / MIN_PB_SIZE);
  int32_t hdr[3] = {0x44444444, e->frame_info.frame_num, cells};
  fwrite(hdr, sizeof hdr, 1, f);
  for (int i = 0; i < cells; i++) {
    const deblock_data_t *d = &e->deblock_data[i];
    int32_t r[14] = {d->mode, d->cbp.y, d->cbp.u, d->cbp.v, d->size, d->tb_split, d->pb_part, d->inter_pred.mv0.x, d->inter_pred.mv0.y,
                     d->inter_pred.mv1.x, d->inter_pred.mv1.y, d->inter_pred.ref_idx0, d->inter_pred.ref_idx1, d->inter_pred.bipred_flag};
    fwrite(r, sizeof r, 1, f);
  }
  fclose(f);
}

typedef void (*enc_fn)(encoder_info_t *);
#ifndef DD_STATIC
static enc_fn next_impl(const char *name) {
  enc_fn f = (enc_fn)dlsym(RTLD_NEXT, name);
  if (!f) { fprintf(stderr, "dd_shim: no %s behind the wrapper (library not linked?)\n", name); abort(); }
  return f;
}
#endif
void encode_frame_lbd(encoder_info_t *e) {
#ifdef DD_STATIC
  ref_encode_frame_lbd(e);
#else
  next_impl("encode_frame_lbd")(e);
#endif
  dd_dump(e);
}
void encode_frame_hbd(encoder_info_t *e) {
#ifdef DD_STATIC
  ref_encode_frame_hbd(e);
#else
  next_impl("encode_frame_hbd")(e);
#endif
  dd_dump(e);
}
