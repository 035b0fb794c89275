#!/bin/bash
# round 5, final call D: the default invocation of bench.py, the 1920x1080 configuration in lock step (A/B of the schedule at that geometry) and the
# headline workload with 192 streams per GPU (what more streams buy: DESIGN.md 9.3).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
line() { echo "$(el) $1: $(grep -o '"value": [0-9.]*' $2 | head -1) $(grep -o '"ms_per_step": [0-9.]*' $2) $(grep -o '"bit_exact": [a-z]*' $2) $(grep -o '"avg_launch_ms": [0-9.]*' $2) $(grep -o '"cpu_baseline": {"value": [0-9.a-z]*' $2)"; }
timeout 300 python bench.py > $O/r05_bench_default.json 2> $O/r05_bench_default.err; line "default invocation" $O/r05_bench_default.json; tail -2 $O/r05_bench_default.err
timeout 300 python bench.py --width 1920 --height 1080 --streams 256 --warmup 5 --steps 8 --lockstep --no-cpu-baseline > $O/r05_bench_1080p_ldb_lockstep.json 2> $O/r05_bench_1080p_ldb_lockstep.err; line "cfg 2 in lock step" $O/r05_bench_1080p_ldb_lockstep.json; tail -2 $O/r05_bench_1080p_ldb_lockstep.err
timeout 600 python bench.py --streams 192 --warmup 5 --steps 20 --no-cpu-baseline > $O/r05_bench_4k_ldb_s192.json 2> $O/r05_bench_4k_ldb_s192.err; line "headline workload with 192 streams" $O/r05_bench_4k_ldb_s192.json; tail -2 $O/r05_bench_4k_ldb_s192.err
