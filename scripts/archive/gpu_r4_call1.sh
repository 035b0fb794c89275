#!/bin/bash
# round 4, call 1: HEAD on the MI355X for the first time (the eight end-of-round-3 commits + the round-4 fixes) -
#   parity of HEAD, parity + speed of the laggards-first queue, A/B r3final (1fc3cf8) vs HEAD, how the throughput depends on the
#   number of resident workgroups (THOR_HIP_WGS), the 2-waves-per-SIMD builds (256 VGPRs: no register-pressure spills; with 32x32
#   blocks in LDS), phase profile of HEAD.
# Built in the container beforehand: libthor_hip.so (HEAD), _r3final (scripts/build_at_commit.sh 1fc3cf8 r3final), _occ2 (-DTK_OCC=2),
#   _occ2l32 (-DTK_OCC=2 -DTK_LDSBLK=32), _prof (-DTHOR_PROF).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$R/gpurun_out
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s]"; }
par() {  # tag lib k-filter timeout [env]
  THOR_HIP_LIB=$2 timeout $4 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_kat.py -q -x -m gpu -k "$3" > $O/r4c1_par_$1.log 2>&1
  echo "$(el) parity $1 rc=$? $(tail -1 $O/r4c1_par_$1.log)"
}
ab() {   # tag lib [extra env assignments...]
  tag=$1; lib=$2; shift 2
  [ -f $lib ] || { echo "ab $tag: $lib missing"; return; }
  env THOR_HIP_LIB=$lib "$@" timeout 300 python bench.py --width 1920 --height 1080 --streams 128 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > $O/r4c1_ab_$tag.log 2>&1
  echo "$(el) ab $tag: $(grep -o '"value": [0-9.]*' $O/r4c1_ab_$tag.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r4c1_ab_$tag.log) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r4c1_ab_$tag.log)"
}
L=$R/thor_amd
timeout 900 python -m pytest tests -q -x -m gpu --durations=8 > $O/r4c1_suite_head.log 2>&1; echo "$(el) full -m gpu suite on HEAD rc=$? $(tail -1 $O/r4c1_suite_head.log)"
ab head $L/libthor_hip.so
ab r3final $L/libthor_hip_r3final.so
ab head_wgs512 $L/libthor_hip.so THOR_HIP_WGS=512
ab head_wgs384 $L/libthor_hip.so THOR_HIP_WGS=384
par occ2 $L/libthor_hip_occ2.so "gpu_matches or two_streams" 300
ab occ2 $L/libthor_hip_occ2.so
par occ2l32 $L/libthor_hip_occ2l32.so "gpu_matches or two_streams" 300
ab occ2l32 $L/libthor_hip_occ2l32.so
# queue discipline of the superblock scheduler (tk_sched.h): parity with laggards first, then the A/B where it matters (4K, 128 streams)
THOR_SCHED=lag par lag $L/libthor_hip.so "gpu_matches or two_streams or six_frames" 400
for q in fifo lag; do
  THOR_SCHED=$q timeout 400 python bench.py --warmup 2 --steps 2 --no-verify --no-cpu-baseline > $O/r4c1_sched_$q.log 2>&1
  echo "$(el) sched $q: $(grep -o '"value": [0-9.]*' $O/r4c1_sched_$q.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r4c1_sched_$q.log)"
done
# phase profile of HEAD (the shares in DESIGN 8 predate the round-3 end changes)
if [ -f $L/libthor_hip_prof.so ]; then
  mkdir -p /tmp/w; python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 7 2
  gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_prof tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_prof.so -Wl,-rpath,$R/thor_amd
  THOR_PROF=1 timeout 300 /tmp/w/thorenc_prof -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 6 -streams 128 -wrap 7 > $O/r4c1_prof.log 2>&1
  echo "$(el) prof rc=$?"; grep -v "^[WIE]2026" $O/r4c1_prof.log | tail -40
fi
