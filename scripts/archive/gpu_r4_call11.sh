#!/bin/bash
# round 4, call 11 (experiment for the next round; variant built from branch r5-prep: CDEF search without the private 64-entry array,
# k_cdef with launch bounds 256): parity of the small goldens (all have cdef 2) and A/B of the filter time.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out
O=$R/gpurun_out
THOR_HIP_LIB=$R/thor_amd/libthor_hip_prep.so timeout 200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "gpu_matches or two_streams" > $O/r4c11_par.log 2>&1; echo "parity prep rc=$? $(tail -1 $O/r4c11_par.log)"
for v in final prep; do
  lib=$R/thor_amd/libthor_hip_$v.so; [ $v = final ] && lib=$R/thor_amd/libthor_hip.so
  THOR_HIP_LIB=$lib timeout 300 python bench.py --width 1920 --height 1080 --streams 256 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > $O/r4c11_ab_$v.log 2>&1
  echo "ab $v s256: $(grep -o '"value": [0-9.]*' $O/r4c11_ab_$v.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r4c11_ab_$v.log) $(grep -o 'filters+ref kernels took [0-9.]* ms' $O/r4c11_ab_$v.log)"
done
