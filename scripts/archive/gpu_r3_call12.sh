#!/bin/bash
# round 3, call 12: SSD sums in registers (one reduction per cost), CfL sums in registers, early-skip flags by ballot (b1): parity, A/B
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$R/gpurun_out
export THOR_HIP_LIB=$R/thor_amd/libthor_hip_b1.so
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "golden or two_streams" > $O/r3c12_par_small.log 2>&1; echo "parity small rc=$? $(tail -1 $O/r3c12_par_small.log)"
timeout 300 python -m pytest tests/test_gpu_kat.py -q -x -m gpu > $O/r3c12_kat.log 2>&1; echo "kat rc=$? $(tail -1 $O/r3c12_kat.log)"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "1080p_ldb_n5 or six_frames" > $O/r3c12_par_big.log 2>&1; echo "parity big rc=$? $(tail -1 $O/r3c12_par_big.log)"
ab() {
  tag=$1
  THOR_HIP_LIB=$R/thor_amd/libthor_hip_$tag.so timeout 300 python bench.py --width 1920 --height 1080 --streams 128 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > $O/r3c12_ab_$tag.log 2>&1
  echo "ab $tag: $(grep -o '"value": [0-9.]*' $O/r3c12_ab_$tag.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r3c12_ab_$tag.log)"
}
ab fin; ab b1; ab fin; ab b1
