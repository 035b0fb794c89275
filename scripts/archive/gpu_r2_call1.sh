#!/bin/bash
# round 2, call 1: the round-1 engine (one wavefront per superblock) on the north-star geometry - 4K parity + a first 4K bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "4k_ldb_n2" ) > gpurun_out/r2c1_tests.log 2>&1
( time timeout 900 python bench.py --streams 96 --steps 2 --warmup 1 ) > gpurun_out/r2c1_bench.log 2>&1
tail -3 gpurun_out/r2c1_tests.log; tail -5 gpurun_out/r2c1_bench.log
