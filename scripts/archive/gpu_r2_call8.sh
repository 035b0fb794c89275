#!/bin/bash
# round 2, call 8: the driver's regime in short form - warm-up 5 (I + 4 P), ONE timed P frame with 4 references; exercises the
# two-frame verification (I + P against the live reference) and the 6-frame cpu_baseline leg of bench.py
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python bench.py --gpus 1 --steps 1 --warmup 5 ) > gpurun_out/r2c8_bench_w5.log 2>&1
grep -v "^[WIE]2026" gpurun_out/r2c8_bench_w5.log | tail -5
