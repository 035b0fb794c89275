/* thorenc_hip.c - minimal C front end over the libthor_hip.so sequence API (include/thor_hip.h).
 * Usage mirrors the reference Thorenc (enc/strings.c:287-356) for the options this path honours:
 *   thorenc_hip -cf config.txt -if in.yuv -width W -height H -qp Q -n N [-skip K] [-f fps]
 *               [-of str.bit] [-rf rec.yuv] [-streams S] [-name value ...]
 * With -streams S > 1, stream s encodes frames [skip + s*N, skip + (s+1)*N) as its own closed
 * stream and writes <of>.<s> / <rf>.<s> (the reference's -skip/-n chunking).
 * All input frames are staged in HBM first; the timed region covers the encode loop only. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "../include/thor_hip.h"

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char** argv) {
  thor_hip_params p;
  const char *inf = NULL, *of = NULL, *rf = NULL;
  int n = 600, skip = 0, S = 1, i, wrap = 0;
  thor_hip_params_from_config(&p, NULL);
  /* config files first, explicit options afterwards (same precedence as the reference) */
  for (i = 1; i + 1 < argc; i += 2)
    if (!strcmp(argv[i], "-cf") && thor_hip_params_from_config(&p, argv[i + 1])) return 2;
  for (i = 1; i + 1 < argc; i += 2) {
    const char *k = argv[i], *v = argv[i + 1];
    if (!strcmp(k, "-cf")) continue;
    else if (!strcmp(k, "-if")) inf = v;
    else if (!strcmp(k, "-of")) of = v;
    else if (!strcmp(k, "-rf")) rf = v;
    else if (!strcmp(k, "-n")) n = atoi(v);
    else if (!strcmp(k, "-skip")) skip = atoi(v);
    else if (!strcmp(k, "-streams")) S = atoi(v);
    else if (!strcmp(k, "-wrap")) wrap = atoi(v); /* clip length: frame index taken modulo this (throughput tests) */
    else if (thor_hip_params_set(&p, k, v)) { fprintf(stderr, "Run-time error...\noption %s %s is unknown or not implemented by this path\n...now exiting to system...\n", k, v); return 2; }
  }
  if (!inf) { fprintf(stderr, "usage: %s -cf cfg -if in.yuv -width W -height H -qp Q -n N ...\n", argv[0]); return 2; }
  FILE* fi = fopen(inf, "rb");
  if (!fi) { fprintf(stderr, "cannot open %s\n", inf); return 2; }
  size_t fsz = (size_t)p.width * p.height * 3 / 2 * (p.bitdepth > 8 ? 2 : 1);
  unsigned char* frame = (unsigned char*)malloc(fsz);
  thor_hip_encoder* e = thor_hip_open(&p, S, 0);
  if (!e) { fprintf(stderr, "thor_hip_open failed\n"); return 3; }
  for (int s = 0; s < S; s++)
    for (int f = 0; f < n; f++) {
      size_t idx = (size_t)skip + (size_t)s * n + f;
      if (wrap > 0) idx %= (size_t)wrap;
      if (fseek(fi, (long)(idx * fsz), SEEK_SET) || fread(frame, 1, fsz, fi) != fsz) { fprintf(stderr, "short read at frame %zu\n", idx); return 4; }
      thor_hip_stage_frame(e, s, f, frame);
    }
  FILE** fr = (FILE**)calloc(S, sizeof(FILE*));
  char name[4096];
  for (int s = 0; s < S && rf; s++) {
    if (S == 1) snprintf(name, sizeof name, "%s", rf); else snprintf(name, sizeof name, "%s.%d", rf, s);
    fr[s] = fopen(name, "wb");
  }
  int* slots = (int*)malloc(S * sizeof(int));
  /* coding-order schedule (B frames are coded out of display order); recon is written in display order */
  fseek(fi, 0, SEEK_END);
  long file_frames = wrap > 0 ? (long)skip + (long)S * n : (long)(ftell(fi) / (long)fsz);
  for (int s = 0; s < S; s++)
    if (thor_hip_begin_sequence(e, s, skip + s * n, n, (int)file_frames)) { fprintf(stderr, "bad sequence bounds\n"); return 5; }
  unsigned char** recs = (unsigned char**)calloc((size_t)S * n, sizeof(unsigned char*));
  double t0 = now_s(), tenc = 0;
  int coded = 0;
  for (;;) {
    int active = 0;
    for (int s = 0; s < S; s++) active += thor_hip_next_frame(e, s, &slots[s]);
    if (!active) break;
    if (active != S) { fprintf(stderr, "streams out of step\n"); return 5; }
    double a = now_s();
    if (thor_hip_encode_staged(e, slots)) { fprintf(stderr, "encode failed\n"); return 5; }
    tenc += now_s() - a;
    coded++;
    for (int s = 0; s < S; s++)
      if (fr[s]) {
        unsigned char* r = (unsigned char*)malloc(fsz);
        thor_hip_get_recon(e, s, r);
        recs[(size_t)s * n + slots[s]] = r;
      }
  }
  for (int s = 0; s < S; s++)
    if (fr[s])
      for (int f = 0; f < n; f++)
        if (recs[(size_t)s * n + f]) fwrite(recs[(size_t)s * n + f], 1, fsz, fr[s]);
  n = coded;
  double sb_ms = 0, filt_ms = 0; long launches = 0;
  thor_hip_kernel_time(e, &sb_ms, &launches, &filt_ms);
  double mpx = (double)p.width * p.height * n * S / 1e6;
  fprintf(stdout, "thorenc_hip: %d stream(s) x %d frame(s) %dx%d: encode %.3f s (%.3f Mpx/s), total %.3f s; superblock kernels %.1f ms in %ld launches, filters %.1f ms\n",
          S, n, p.width, p.height, tenc, mpx / tenc, now_s() - t0, sb_ms, launches, filt_ms);
  if (getenv("THOR_PROF")) {
    static const char* nm[32] = {"sb_total", "early_skip", "me_fullpel", "me_subpel", "pred_inter", "md_worker_all_waves", "code_tu", "bits", "cost", "final", "subpel_loop",
                                 "me_calls(n)", "quant", "me_telescope", "me_cands", "me_hex", "tu4", "tu8", "tu16", "tu32", "tu64+",
                                 "tu4(n)", "tu8(n)", "tu16(n)", "tu32(n)", "tu64+(n)", "wg_barrier_wait(all waves)", "bipred_lockstep(all waves, incl. barriers)", "helpers_parked(master alone)", "md_fork_to_join(master)", "tu_fwd", "tu_inv"};
    /* THOR_PROF=md: the library was built with -DTHOR_PROF_MD (slots 16..25 = work-queue items by kind / master-alone phases, tk_block.h);
       THOR_PROF=me: -DTHOR_PROF_ME (motion-search cycles and calls by coding-block size) */
    static const char* nm_md[10] = {"items skip/merge", "items intra", "items search (MD_REF)", "items trial (incl. wait)", "trial items: wait for vectors", "queue set-up (master)",
                                    "block entry (master)", "early-skip path (master)", "final encode of decided blocks: emission (master)", "final encode of decided blocks: whole (master)"};
    static const char* nm_me[10] = {"me cb8", "me cb16", "me cb32", "me cb64", "me cb128", "me cb8(n)", "me cb16(n)", "me cb32(n)", "me cb64(n)", "me cb128(n)"};
    if (!strcmp(getenv("THOR_PROF"), "md")) for (int k = 0; k < 10; k++) nm[16 + k] = nm_md[k];
    if (!strcmp(getenv("THOR_PROF"), "me")) for (int k = 0; k < 10; k++) nm[16 + k] = nm_me[k];
    long long pr[32];
    thor_hip_read_prof(e, pr);
    for (int k = 0; k < 32; k++) fprintf(stdout, "prof %-40s %16lld %6.2f%%\n", nm[k], pr[k], pr[0] ? 100.0 * pr[k] / pr[0] : 0.0);
  }
  for (int s = 0; s < S; s++) {
    if (fr[s]) fclose(fr[s]);
    if (of) {
      if (S == 1) snprintf(name, sizeof name, "%s", of); else snprintf(name, sizeof name, "%s.%d", of, s);
      FILE* fo = fopen(name, "wb");
      fwrite(thor_hip_stream_data(e, s), 1, thor_hip_stream_bytes(e, s), fo);
      fclose(fo);
    }
  }
  thor_hip_close(e);
  fclose(fi);
  return 0;
}
