#!/bin/bash
# round 4, call 4: the candidate final library (search windows for PUs up to 32x32 / 16-bit, 16-bit dot-product sub-pel incl. strips,
# joint-search item of B frames, cooperative emission, vector sample loops incl. transform residual / reconstruction, 256-VGPR budget
# of the 16-bit instances): full -m gpu suite, A/B against HEAD of call 1 (libthor_hip_head.so = de0fe67) and against the same
# sources without the vector loops (-DTK_NOVEC), the RA / HDB16 operating points, a verified 3840x2160 run in the driver's regime.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$R/gpurun_out
L=$R/thor_amd
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s]"; }
timeout 1000 python -m pytest tests -q -x -m gpu --durations=6 > $O/r4c4_suite.log 2>&1; echo "$(el) full -m gpu suite rc=$? $(tail -1 $O/r4c4_suite.log)"
ab() {   # tag lib streams
  THOR_HIP_LIB=$2 timeout 300 python bench.py --width 1920 --height 1080 --streams $3 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > $O/r4c4_ab_$1.log 2>&1
  echo "$(el) ab $1: $(grep -o '"value": [0-9.]*' $O/r4c4_ab_$1.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r4c4_ab_$1.log) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r4c4_ab_$1.log)"
}
ab head_s128 $L/libthor_hip_head.so 128
ab new_s128 $L/libthor_hip.so 128
ab novec_s128 $L/libthor_hip_novec.so 128
ab head_s256 $L/libthor_hip_head.so 256
ab new_s256 $L/libthor_hip.so 256
timeout 300 python bench.py --config ra --width 1920 --height 1080 --streams 96 --warmup 1 --steps 8 --no-verify --no-cpu-baseline > $O/r4c4_ra_1080p.log 2>&1
echo "$(el) ra 1080p s96: $(grep -o '"value": [0-9.]*' $O/r4c4_ra_1080p.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r4c4_ra_1080p.log)"
timeout 400 python bench.py --config hdb16 --bitdepth 10 --width 1920 --height 1080 --streams 96 --warmup 1 --steps 16 --no-verify --no-cpu-baseline > $O/r4c4_hdb16_1080p.log 2>&1
echo "$(el) hdb16 10-bit 1080p s96: $(grep -o '"value": [0-9.]*' $O/r4c4_hdb16_1080p.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r4c4_hdb16_1080p.log)"
timeout 600 python bench.py --warmup 5 --steps 8 > $O/r4c4_bench_4k.log 2> $O/r4c4_bench_4k.err
echo "$(el) 4K driver regime (8 timed frames), verified: $(grep -o '"value": [0-9.]*' $O/r4c4_bench_4k.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r4c4_bench_4k.log) $(grep -o '"bit_exact": [a-z]*' $O/r4c4_bench_4k.log) $(grep -o '"cpu_baseline": {"value": [0-9.a-z]*' $O/r4c4_bench_4k.log)"; tail -3 $O/r4c4_bench_4k.err
