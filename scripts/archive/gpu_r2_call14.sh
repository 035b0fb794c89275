#!/bin/bash
# round 2, call 14: default bench line of the final default library (search window off)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 160 python bench.py > gpurun_out/r2c14_bench.log 2>&1
grep -v "^[WIE]2026" gpurun_out/r2c14_bench.log | tail -2
