#!/bin/bash
# round 5, call 5: which of the second-round motion-search changes cost time (call 4: +10 % per call): A = taps as immediates + 24-bit multiplies,
# C = A + the rate term formed with the candidate, D = C + the rate term from the LDS table (a wave vote + branch per candidate); full-pel part of A alone.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; O=$R/gpurun_out
for b in ubench_me_r1 ubench_me_A ubench_me_C ubench_me_D ubench_me_nosubA; do timeout 120 tools/$b > $O/r5c5_$b.log 2>&1; echo "== $b"; grep "1 workgroup" $O/r5c5_$b.log; done
