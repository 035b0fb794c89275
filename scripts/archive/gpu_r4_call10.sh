#!/bin/bash
# round 4, call 10 (planning data, final library): what a fuller chip is worth at 3840x2160 - 192 instead of 128 streams, frames P5 / P6,
# with THOR_SBTIMES stamps (upper bound of what un-locking the streams' frames from each other can recover).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out /tmp/w
O=$R/gpurun_out
THOR_SBTIMES=/tmp/w/sbt192.bin timeout 500 python bench.py --streams 192 --warmup 5 --steps 2 --no-verify --no-cpu-baseline > $O/r4c10_s192.log 2>&1
echo "s192: $(grep -o '"value": [0-9.]*' $O/r4c10_s192.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r4c10_s192.log)"; python scripts/sbtimes.py /tmp/w/sbt192.bin 768 > $O/r4c10_sbtimes_s192.log 2>&1; tail -4 $O/r4c10_sbtimes_s192.log
