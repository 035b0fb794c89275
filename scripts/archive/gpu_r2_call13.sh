#!/bin/bash
# round 2, call 13 (last): LDS search window A/B, then the full -m gpu suite and the default bench line on the final library
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
ab() {
  tag=$1; lib=$2; shift 2
  THOR_HIP_LIB=$R/thor_amd/$lib timeout 200 python bench.py --width 1920 --height 1080 --streams 128 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > gpurun_out/r2c13_ab_$tag.log 2>&1
  echo "$tag: $(grep -o '"value": [0-9.]*' gpurun_out/r2c13_ab_$tag.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2c13_ab_$tag.log)"
}
ab nowin libthor_hip_nowin.so
ab win libthor_hip.so
( time timeout 600 python -m pytest tests -m gpu -q -x --durations=4 ) > gpurun_out/r2c13_tests.log 2>&1
tail -9 gpurun_out/r2c13_tests.log
( time timeout 400 python bench.py ) > gpurun_out/r2c13_bench.log 2>&1
grep -v "^[WIE]2026" gpurun_out/r2c13_bench.log | tail -4
