#!/bin/bash
# round 6, call 10: (a) master placement - the roles of a workgroup's wavefronts rotated so that the masters of the workgroups of one CU sit on different
# SIMDs (tk_kernel.h; THOR_HIP_MASTER=wave0 = the old behaviour: physical wavefront 0 is the master): where the hardware puts the wavefronts (THOR_SIMDMAP),
# parity, A/B; (b) the same sources built with other instruction-scheduling strategies of the compiler (thor_amd/libthor_hip_{ilp,memcl,trk,bias0}.so).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
AB="--width 1920 --height 1080 --streams 256 --warmup 5 --steps 4 --no-cpu-baseline"
THOR_SIMDMAP=1 timeout 300 python bench.py --width 1920 --height 1080 --streams 256 --warmup 0 --steps 1 --no-cpu-baseline --no-verify > $O/r6c10_map.log 2> $O/r6c10_map.err
echo "$(el) $(grep 'simd map' $O/r6c10_map.err | head -2)"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/r6c10_parity.log 2>&1; echo "$(el) parity (master placement on) rc=$? $(tail -1 $O/r6c10_parity.log)"; grep -E "^FAILED|^ERROR" $O/r6c10_parity.log | head
for v in wave0 spread wave0 spread; do
  m=; [ $v = wave0 ] && m=wave0
  THOR_HIP_MASTER=$m timeout 400 python bench.py $AB > $O/r6c10_ab_$v.log 2>$O/r6c10_ab_$v.err
  echo "$(el) 1080p s256 P5-P8 $v: $(grep -o '"value": [0-9.]*' $O/r6c10_ab_$v.log | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r6c10_ab_$v.log) $(grep -o '"k_superblocks": [0-9.]*' $O/r6c10_ab_$v.log | head -1)"
done
for v in ilp memcl trk bias0; do
  [ -f $R/thor_amd/libthor_hip_$v.so ] || continue
  THOR_HIP_LIB=$R/thor_amd/libthor_hip_$v.so timeout 400 python bench.py $AB > $O/r6c10_flag_$v.log 2>$O/r6c10_flag_$v.err
  echo "$(el) 1080p s256 P5-P8 build $v: $(grep -o '"value": [0-9.]*' $O/r6c10_flag_$v.log | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r6c10_flag_$v.log) $(grep -o '"k_superblocks": [0-9.]*' $O/r6c10_flag_$v.log | head -1) $(grep -o '"superblock_kernel": {[^}]*}' $O/r6c10_flag_$v.log)"
done
