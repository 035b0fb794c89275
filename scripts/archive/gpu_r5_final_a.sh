#!/bin/bash
# round 5, final call A: everything that is reported about the library that ships, part 1.
#   1. full -m gpu suite
#   2. the driver's regime: python bench.py --steps 20 --warmup 5 (self-verifying: live reference runs + recorded reference runs of every timed frame; cpu_baseline)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
line() { echo "$(el) $1: $(grep -o '"value": [0-9.]*' $2 | head -1) $(grep -o '"ms_per_step": [0-9.]*' $2) $(grep -o '"bit_exact": [a-z]*' $2) $(grep -o '"avg_launch_ms": [0-9.]*' $2) $(grep -o '"cpu_baseline": {"value": [0-9.a-z]*' $2)"; }
timeout 1200 python -m pytest tests -q -x -m gpu --durations=8 > $O/r5f_suite.log 2>&1; echo "$(el) full -m gpu suite rc=$? $(tail -1 $O/r5f_suite.log)"; grep -E "^[0-9.]+s " $O/r5f_suite.log | head -8
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r05_bench_driver_regime.json 2> $O/r05_bench_driver_regime.err; line "driver regime" $O/r05_bench_driver_regime.json; tail -3 $O/r05_bench_driver_regime.err
