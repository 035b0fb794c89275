#!/bin/bash
# round 5, call 3: motion_estimate in isolation (tools/ubench_me: cycles per call by PU size and CU residency, before / after the select-based
# vector-bit count + 32-bit cost keys) and the same change in the encoder (A/B on the driver-regime proxy, lock step to keep the schedule out of it).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; O=$R/gpurun_out
timeout 120 tools/ubench_me_old > $O/r5c3_ubench_me_old.log 2>&1; cat $O/r5c3_ubench_me_old.log
timeout 120 tools/ubench_me > $O/r5c3_ubench_me_new.log 2>&1; cat $O/r5c3_ubench_me_new.log
timeout 200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "gpu_matches and (n3_q32 or n6_q24 or n27 or 10bit or ra)" > $O/r5c3_par.log 2>&1; echo "parity new rc=$? $(tail -1 $O/r5c3_par.log)"
for v in base5 new; do
  lib=$R/thor_amd/libthor_hip_$v.so; [ $v = new ] && lib=$R/thor_amd/libthor_hip.so
  THOR_HIP_LIB=$lib timeout 300 python bench.py --width 1920 --height 1080 --streams 256 --warmup 5 --steps 4 --no-verify --no-cpu-baseline --lockstep > $O/r5c3_ab_$v.log 2>$O/r5c3_ab_$v.err
  echo "1080p s256 P5-P8 lockstep $v: $(grep -o '"value": [0-9.]*' $O/r5c3_ab_$v.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r5c3_ab_$v.log)"
done
