#!/bin/bash
# round 5, call 9: large-PU evaluator (candidates formed once per lane + v_readlane, register-held original segments, two candidates in flight for
# 64x64): parity incl. the 3840x2160 goldens, A/B.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; O=$R/gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "gpu_matches" > $O/r5c9_par.log 2>&1; echo "parity rc=$? $(tail -1 $O/r5c9_par.log)"
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "4k_ldb_n6 or 4k_ra or 1080p_ldb_n5 or two_frames or each_equals" > $O/r5c9_par_big.log 2>&1; echo "parity big rc=$? $(tail -1 $O/r5c9_par_big.log)"
for v in cand3 new; do
  lib=$R/thor_amd/libthor_hip_$v.so; [ $v = new ] && lib=$R/thor_amd/libthor_hip.so
  THOR_HIP_LIB=$lib timeout 300 python bench.py --width 1920 --height 1080 --streams 256 --warmup 5 --steps 4 --no-verify --no-cpu-baseline --lockstep > $O/r5c9_ab_$v.log 2>$O/r5c9_ab_$v.err
  echo "1080p s256 P5-P8 lockstep $v: $(grep -o '"value": [0-9.]*' $O/r5c9_ab_$v.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r5c9_ab_$v.log)"
done
