#!/bin/bash
# round 5, call 6: the one-lane-per-candidate motion search (me_cand8_fullpel / me_cand8_subpel) in the encoder - parity with the cross-check
# library (both searches run, the kernel traps when they disagree) and with the product library, then A/B on the driver-regime proxy.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; O=$R/gpurun_out
THOR_HIP_LIB=$R/thor_amd/libthor_hip_xcheck.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "gpu_matches or staggered or two_streams" > $O/r5c6_par_xcheck.log 2>&1; echo "parity xcheck rc=$? $(tail -1 $O/r5c6_par_xcheck.log)"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu > $O/r5c6_par.log 2>&1; echo "parity new rc=$? $(tail -1 $O/r5c6_par.log)"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "six_frames or 1080p_ldb_n5 or 4k_ldb_n2" > $O/r5c6_par_big.log 2>&1; echo "parity big rc=$? $(tail -1 $O/r5c6_par_big.log)"
for v in base5 new; do
  lib=$R/thor_amd/libthor_hip_$v.so; [ $v = new ] && lib=$R/thor_amd/libthor_hip.so
  THOR_HIP_LIB=$lib timeout 300 python bench.py --width 1920 --height 1080 --streams 256 --warmup 5 --steps 4 --no-verify --no-cpu-baseline --lockstep > $O/r5c6_ab_$v.log 2>$O/r5c6_ab_$v.err
  echo "1080p s256 P5-P8 lockstep $v: $(grep -o '"value": [0-9.]*' $O/r5c6_ab_$v.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r5c6_ab_$v.log)"
done
