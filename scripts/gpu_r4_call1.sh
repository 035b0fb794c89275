#!/bin/bash
# round 4, call 1 (prepared at the end of round 3, whose GPU budget was spent before these changes existed): parity and A/B of the
# exact work-avoidance / scheduling steps that were verified in the host simulations only -
#   r3final = 1fc3cf8 (last library measured on the MI355X: 93.2 Mpx/s driver regime), es = + early-skip reuse (29ce444),
#   split = + search / trial queue items, dd = + no vector evaluated twice (telescope / hexagon), hb = + header bits in every pruning
#   bound, head = + candidate-list compaction (HEAD).
# Build the variants in the container first (the libraries travel with the snapshot):
#   scripts/build_at_commit.sh 1fc3cf8 r3final; scripts/build_at_commit.sh 29ce444 es; scripts/build_at_commit.sh bf8d06b split; scripts/build_at_commit.sh 395596c dd; scripts/build_at_commit.sh 8062e71 hb
#   scripts/build_variant.sh prof -DTHOR_PROF        (phase profile of HEAD, last step of this script)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$R/gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "golden or two_streams" > $O/r4c1_par_small.log 2>&1; echo "parity small rc=$? $(tail -1 $O/r4c1_par_small.log)"
timeout 700 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "1080p_ldb_n5 or six_frames or ra" > $O/r4c1_par_big.log 2>&1; echo "parity big rc=$? $(tail -1 $O/r4c1_par_big.log)"
ab() {
  tag=$1; lib=$R/thor_amd/libthor_hip_$tag.so; [ "$tag" = head ] && lib=$R/thor_amd/libthor_hip.so
  [ -f $lib ] || { echo "ab $tag: $lib missing"; return; }
  THOR_HIP_LIB=$lib timeout 300 python bench.py --width 1920 --height 1080 --streams 128 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > $O/r4c1_ab_$tag.log 2>&1
  echo "ab $tag: $(grep -o '"value": [0-9.]*' $O/r4c1_ab_$tag.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r4c1_ab_$tag.log)"
}
ab r3final; ab es; ab split; ab dd; ab hb; ab head
abra() {
  tag=$1; lib=$R/thor_amd/libthor_hip_$tag.so; [ "$tag" = head ] && lib=$R/thor_amd/libthor_hip.so
  [ -f $lib ] || return
  THOR_HIP_LIB=$lib timeout 400 python bench.py --config ra --width 1920 --height 1080 --streams 96 --warmup 1 --steps 8 --no-verify --no-cpu-baseline > $O/r4c1_abra_$tag.log 2>&1
  echo "ab RA $tag: $(grep -o '"value": [0-9.]*' $O/r4c1_abra_$tag.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r4c1_abra_$tag.log)"
}
abra r3final; abra head
# queue discipline of the superblock scheduler (tk_sched.h): parity with laggards first, then the A/B where it matters (4K, 128 streams)
THOR_SCHED=lag timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -m gpu -k "golden or two_streams or six_frames" > $O/r4c1_par_lag.log 2>&1; echo "parity lag rc=$? $(tail -1 $O/r4c1_par_lag.log)"
for q in fifo lag; do
  THOR_SCHED=$q timeout 500 python bench.py --warmup 5 --steps 2 --no-verify --no-cpu-baseline > $O/r4c1_sched_$q.log 2>&1
  echo "sched $q: $(grep -o '"value": [0-9.]*' $O/r4c1_sched_$q.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r4c1_sched_$q.log)"
done
# phase profile of HEAD (the shares in DESIGN 8 predate the round-3 end changes)
if [ -f $R/thor_amd/libthor_hip_prof.so ]; then
  mkdir -p /tmp/w; python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 7 2
  gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_prof tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_prof.so -Wl,-rpath,$R/thor_amd
  THOR_PROF=1 timeout 300 /tmp/w/thorenc_prof -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 6 -streams 128 -wrap 7 > $O/r4c1_prof.log 2>&1
  echo "prof rc=$?"; grep -v "^[WIE]2026" $O/r4c1_prof.log | tail -36
fi
