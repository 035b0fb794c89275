#!/bin/bash
# round 5, call 4: second round of motion-search instruction work (rate term from an LDS table formed with the candidate, 24-bit multiplies, filter taps
# as immediates instead of constant-memory tables, packed taps straight into the dot-product form): ubench per call, full-pel part alone, A/B in the encoder.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; O=$R/gpurun_out
for b in ubench_me_r1 ubench_me ubench_me_nosub; do timeout 120 tools/$b > $O/r5c4_$b.log 2>&1; echo "== $b"; cat $O/r5c4_$b.log; done
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "gpu_matches or staggered" > $O/r5c4_par.log 2>&1; echo "parity new rc=$? $(tail -1 $O/r5c4_par.log)"
for v in base5 new; do
  lib=$R/thor_amd/libthor_hip_$v.so; [ $v = new ] && lib=$R/thor_amd/libthor_hip.so
  THOR_HIP_LIB=$lib timeout 300 python bench.py --width 1920 --height 1080 --streams 256 --warmup 5 --steps 4 --no-verify --no-cpu-baseline --lockstep > $O/r5c4_ab_$v.log 2>$O/r5c4_ab_$v.err
  echo "1080p s256 P5-P8 lockstep $v: $(grep -o '"value": [0-9.]*' $O/r5c4_ab_$v.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r5c4_ab_$v.log)"
done
