#!/bin/bash
# round 2, call 9: original block of small coding blocks in LDS (A/B against the committed default library + parity)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
ab() {
  tag=$1; lib=$2; shift 2
  THOR_HIP_LIB=$R/thor_amd/$lib timeout 300 python bench.py --width 1920 --height 1080 --streams 128 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > gpurun_out/r2c9_ab_$tag.log 2>&1
  echo "$tag: $(grep -o '"value": [0-9.]*' gpurun_out/r2c9_ab_$tag.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2c9_ab_$tag.log)"
}
ab base libthor_hip.so
ab e2 libthor_hip_e2.so
( time THOR_HIP_LIB=$R/thor_amd/libthor_hip_e2.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x ) > gpurun_out/r2c9_tests_e2.log 2>&1
tail -3 gpurun_out/r2c9_tests_e2.log
