#!/bin/bash
# round 4, call 2: per-commit A/B of the end-of-round-3 steps, vectorised sample loops (vec), occupancy variants at a geometry that
# fills the chip (1080p x 256 streams), work-queue profile (THOR_PROF_MD) of the LDB and the RA operating points, workgroup
# utilisation at 3840x2160 x 128 streams for both queue disciplines, baselines of the RA / HDB16 operating points.
# Built beforehand: libthor_hip_{vec,es,split,dd,hb,occ2,occ2l32,prof,profmd}.so (scripts/build_variant.sh, scripts/build_at_commit.sh).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
O=$R/gpurun_out
L=$R/thor_amd
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s]"; }
par() {  # tag lib k-filter timeout
  THOR_HIP_LIB=$2 timeout $4 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_kat.py -q -x -m gpu -k "$3" > $O/r4c2_par_$1.log 2>&1
  echo "$(el) parity $1 rc=$? $(tail -1 $O/r4c2_par_$1.log)"
}
ab() {   # tag lib streams [extra env assignments...]
  tag=$1; lib=$2; S=$3; shift 3
  [ -f $lib ] || { echo "ab $tag: $lib missing"; return; }
  env THOR_HIP_LIB=$lib "$@" timeout 300 python bench.py --width 1920 --height 1080 --streams $S --warmup 4 --steps 2 --no-verify --no-cpu-baseline > $O/r4c2_ab_$tag.log 2>&1
  echo "$(el) ab $tag: $(grep -o '"value": [0-9.]*' $O/r4c2_ab_$tag.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r4c2_ab_$tag.log) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r4c2_ab_$tag.log)"
}
par vec $L/libthor_hip_vec.so "gpu_matches or two_streams or six_frames or 1080p_ldb_n5" 400
ab vec $L/libthor_hip_vec.so 128
for v in es split dd hb; do ab $v $L/libthor_hip_$v.so 128; done
ab head_s256 $L/libthor_hip.so 256
ab occ2_s256 $L/libthor_hip_occ2.so 256
ab occ2l32_s256 $L/libthor_hip_occ2l32.so 256
# profiles through the C front end
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 9 2
prof() {  # tag lib mode cfg n streams
  gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_$1 tools/thorenc_hip.c -Lthor_amd -l:$(basename $2) -Wl,-rpath,$R/thor_amd
  THOR_PROF=$3 timeout 300 /tmp/w/thorenc_$1 -cf $R/configs/$4 -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp ${7:-32} -f 30 -n $5 -streams $6 -wrap 9 > $O/r4c2_prof_$1.log 2>&1
  echo "$(el) prof $1 rc=$?"; grep -v "^[WIE]2026" $O/r4c2_prof_$1.log | grep -E "thorenc_hip:|items|master|sb_total|barrier|parked|fork|lockstep|me_fullpel|me_subpel|code_tu|pred_inter"
}
prof ldb_md $L/libthor_hip_profmd.so md ldb_high_efficiency.cfg 6 128
prof ra $L/libthor_hip_prof.so 1 ra_high_efficiency.cfg 9 96 27
prof ra_md $L/libthor_hip_profmd.so md ra_high_efficiency.cfg 9 96 27
# RA / HDB16 operating points, 1080p (A/B baselines of the round) and RA at 3840x2160 with 96 streams
timeout 300 python bench.py --config ra --width 1920 --height 1080 --streams 96 --warmup 1 --steps 8 --no-verify --no-cpu-baseline > $O/r4c2_ra_1080p.log 2>&1
echo "$(el) ra 1080p s96: $(grep -o '"value": [0-9.]*' $O/r4c2_ra_1080p.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r4c2_ra_1080p.log)"
timeout 400 python bench.py --config hdb16 --bitdepth 10 --width 1920 --height 1080 --streams 96 --warmup 1 --steps 16 --no-verify --no-cpu-baseline > $O/r4c2_hdb16_1080p.log 2>&1
echo "$(el) hdb16 10-bit 1080p s96: $(grep -o '"value": [0-9.]*' $O/r4c2_hdb16_1080p.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r4c2_hdb16_1080p.log)"
# workgroup utilisation at 3840x2160, 128 streams, frames P5/P6, both queue disciplines
for q in fifo lag; do
  THOR_SCHED=$q THOR_SBTIMES=/tmp/w/sbt_$q.bin timeout 400 python bench.py --warmup 5 --steps 2 --no-verify --no-cpu-baseline > $O/r4c2_sched_$q.log 2>&1
  echo "$(el) sched $q: $(grep -o '"value": [0-9.]*' $O/r4c2_sched_$q.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r4c2_sched_$q.log)"
  python scripts/sbtimes.py /tmp/w/sbt_$q.bin 768 > $O/r4c2_sbtimes_$q.log 2>&1; tail -4 $O/r4c2_sbtimes_$q.log
done
timeout 500 python bench.py --config ra --streams 96 --warmup 1 --steps 8 --no-verify --no-cpu-baseline > $O/r4c2_ra_4k.log 2>&1
echo "$(el) ra 4K s96: $(grep -o '"value": [0-9.]*' $O/r4c2_ra_4k.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r4c2_ra_4k.log)"
