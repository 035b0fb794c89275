#!/bin/bash
# round 2, call 15 (last seconds of the budget): parity of the final library on the cases that reach 4 references / B frames
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 40 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "10bit or 1080p or dropin_reference or ldb_medium or q44" > gpurun_out/r2c15_tests.log 2>&1
tail -3 gpurun_out/r2c15_tests.log
