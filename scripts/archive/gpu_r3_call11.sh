#!/bin/bash
# round 3, call 11: issue priority for the master wave (A/B, parity subset) + the hard-content line (sigma 6) of the benched configuration
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$R/gpurun_out
THOR_HIP_LIB=$R/thor_amd/libthor_hip_prio2.so timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "golden or two_streams" > $O/r3c11_par_small.log 2>&1; echo "parity small rc=$? $(tail -1 $O/r3c11_par_small.log)"
ab() {
  tag=$1
  THOR_HIP_LIB=$R/thor_amd/libthor_hip_$tag.so timeout 300 python bench.py --width 1920 --height 1080 --streams 128 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > $O/r3c11_ab_$tag.log 2>&1
  echo "ab $tag: $(grep -o '"value": [0-9.]*' $O/r3c11_ab_$tag.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r3c11_ab_$tag.log)"
}
ab fin; ab prio2
timeout 600 python bench.py --sigma 6 --warmup 5 --steps 1 --no-verify --no-cpu-baseline > $O/r3_bench_sigma6.json 2> $O/r3_bench_sigma6.err; echo "sigma6 rc=$?"; cut -c1-300 $O/r3_bench_sigma6.json; grep -o '"content": {[^}]*}' $O/r3_bench_sigma6.json
