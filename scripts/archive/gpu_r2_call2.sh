#!/bin/bash
# round 2, call 2: first run of the 4-waves-per-superblock engine: smoke, parity suites, default bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r2c2_smoke.log 2>&1 || { tail -20 gpurun_out/r2c2_smoke.log; exit 1; }
tail -2 gpurun_out/r2c2_smoke.log
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py -m gpu -x -q ) > gpurun_out/r2c2_tests.log 2>&1
tail -4 gpurun_out/r2c2_tests.log
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --durations=10 ) > gpurun_out/r2c2_fullsize.log 2>&1
tail -16 gpurun_out/r2c2_fullsize.log
( time timeout 600 python bench.py ) > gpurun_out/r2c2_bench.log 2>&1
tail -4 gpurun_out/r2c2_bench.log
