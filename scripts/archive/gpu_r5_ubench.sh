#!/bin/bash
# motion_estimate micro-benchmarks (tools/ubench_me*): every binary in tools/ named ubench_me*, 1 workgroup per CU lines + the phase split of the -DTHOR_PROF build
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; O=$R/gpurun_out
for b in tools/ubench_me*; do [ -x $b ] && [ ! -d $b ] && [[ $b != *.cpp ]] || continue; n=$(basename $b); timeout 120 $b > $O/r5ub_$n.log 2>&1; echo "== $n"; grep "1 workgroup\|per call" $O/r5ub_$n.log; done
