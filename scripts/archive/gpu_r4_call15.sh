#!/bin/bash
# round 4, call 15 (experiment for the next round; variant built from branch r5-prep: one search window per (block, reference) for the
# HOR / VER / QUAD searches; the branch also carries the CDEF variant of call 11, so compare the k_superblocks launch times): parity of the
# small goldens, A/B on the easy frames (P4, P5) and over 13 frames (the later frames are search-bound).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out
O=$R/gpurun_out
THOR_HIP_LIB=$R/thor_amd/libthor_hip_cbwin.so timeout 200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "gpu_matches or two_streams" > $O/r4c15_par.log 2>&1; echo "parity cbwin rc=$? $(tail -1 $O/r4c15_par.log)"
for v in final cbwin; do
  lib=$R/thor_amd/libthor_hip_$v.so; [ $v = final ] && lib=$R/thor_amd/libthor_hip.so
  THOR_HIP_LIB=$lib timeout 300 python bench.py --width 1920 --height 1080 --streams 256 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > $O/r4c15_ab_$v.log 2>&1
  echo "ab $v s256 P4-P5: $(grep -o '"value": [0-9.]*' $O/r4c15_ab_$v.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r4c15_ab_$v.log)"
  THOR_HIP_LIB=$lib timeout 300 python bench.py --width 1920 --height 1080 --streams 128 --warmup 5 --steps 8 --no-verify --no-cpu-baseline > $O/r4c15_ab13_$v.log 2>&1
  echo "ab $v s128 P5-P12: $(grep -o '"value": [0-9.]*' $O/r4c15_ab13_$v.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r4c15_ab13_$v.log)"
done
