#!/bin/bash
# round 6, call 2: what does co-residency cost?  (a) the product library with 256 / 512 / 768 resident workgroups (THOR_HIP_WGS) on the same
# 1080p x 256-stream frames; (b) workgroups of ONE and of TWO wavefronts per superblock (-DTK_WAVES=1: 245 VGPRs, no spills, no parked helpers,
# 8 workgroups per CU; -DTK_WAVES=2: 5 per CU) on 512 streams, with a parity check of each variant.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
AB="--width 1920 --height 1080 --warmup 5 --steps 4 --no-verify --no-cpu-baseline --lockstep"
for n in 256 512 768; do
  THOR_HIP_WGS=$n timeout 400 python bench.py $AB --streams 256 > $O/r6c2_wgs$n.log 2>$O/r6c2_wgs$n.err
  echo "$(el) product, $n resident workgroups, 1080p s256: $(grep -o '"value": [0-9.]*' $O/r6c2_wgs$n.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r6c2_wgs$n.log)"
done
for v in w1 w2; do
  THOR_HIP_LIB=$R/thor_amd/libthor_hip_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "golden or staggered" > $O/r6c2_par_$v.log 2>&1; echo "$(el) parity $v rc=$? $(tail -1 $O/r6c2_par_$v.log)"
done
for v in new w1 w2; do
  lib=$R/thor_amd/libthor_hip_$v.so; [ $v = new ] && lib=$R/thor_amd/libthor_hip.so
  for s in 256 512; do
    THOR_HIP_LIB=$lib timeout 500 python bench.py $AB --streams $s > $O/r6c2_ab_${v}_s$s.log 2>$O/r6c2_ab_${v}_s$s.err
    echo "$(el) 1080p s$s P5-P8 lockstep $v: $(grep -o '"value": [0-9.]*' $O/r6c2_ab_${v}_s$s.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r6c2_ab_${v}_s$s.log)"
  done
done
THOR_HIP_LIB=$R/thor_amd/libthor_hip_w1.so timeout 500 python bench.py $AB --streams 768 > $O/r6c2_ab_w1_s768.log 2>$O/r6c2_ab_w1_s768.err
echo "$(el) 1080p s768 P5-P8 lockstep w1: $(grep -o '"value": [0-9.]*' $O/r6c2_ab_w1_s768.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r6c2_ab_w1_s768.log)"
