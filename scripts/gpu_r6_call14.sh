#!/bin/bash
# round 6, call 14: workgroup utilisation of the steady-state launches (THOR_SBTIMES, scripts/sbtimes.py) with 128 and with 160 streams - what the 3 % of call 12 is.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
for s in 128 160; do
  THOR_SBTIMES=/tmp/w/sbt_$s.bin timeout 240 python bench.py --streams $s --warmup 5 --steps 2 --no-verify --no-cpu-baseline > $O/r6c14_s$s.log 2>&1
  echo "$(el) $s streams: $(grep -o '"value": [0-9.]*' $O/r6c14_s$s.log | head -1)"
  python3 scripts/sbtimes.py /tmp/w/sbt_$s.bin 768 > $O/r6c14_sbtimes_s$s.log 2>&1; tail -6 $O/r6c14_sbtimes_s$s.log | cut -c1-260
done
