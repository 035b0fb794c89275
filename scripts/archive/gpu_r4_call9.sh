#!/bin/bash
# round 4, call 9 (experiment for the next round, scratch variant: profiles/r04_exp_trials_before_intra.diff): the intra items of a block
# decision queued BEHIND the trial items when there are at least as many references as waves - exact (keys), changes only what the pruning catches.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out
O=$R/gpurun_out
for v in final tfirst; do
  lib=$R/thor_amd/libthor_hip_$v.so; [ $v = final ] && lib=$R/thor_amd/libthor_hip.so
  THOR_HIP_LIB=$lib timeout 300 python bench.py --width 1920 --height 1080 --streams 256 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > $O/r4c9_ab_$v.log 2>&1
  echo "ab $v s256: $(grep -o '"value": [0-9.]*' $O/r4c9_ab_$v.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r4c9_ab_$v.log)"
done
THOR_HIP_LIB=$R/thor_amd/libthor_hip_tfirst.so timeout 200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "gpu_matches or two_streams" > $O/r4c9_par.log 2>&1; echo "parity tfirst rc=$? $(tail -1 $O/r4c9_par.log)"
