#!/bin/bash
# round 6, call 13: where the eight-wavefront build stops paying - lat against wide beyond the selection threshold (streams x superblocks in flight > CUs):
# 24 / 32 / 48 streams at 3840x2160 (the library picks lat there), 32 / 64 at 1920x1080.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
line() { echo "$(grep -o '"value": [0-9.]*' $1 | head -1) $(grep -o '"superblock_kernel": {[^}]*}' $1 | cut -c1-70)"; }
for s in 24 32 48; do
  for k in lat wide; do
    THOR_HIP_KERNEL=$k timeout 500 python bench.py --streams $s --warmup 5 --steps 4 --no-verify --no-cpu-baseline > $O/r6c13_4k_s${s}_$k.json 2> $O/r6c13_4k_s${s}_$k.err
    echo "$(el) 4K LDB $s streams forced $k: $(line $O/r6c13_4k_s${s}_$k.json)"
  done
done
for s in 32 64; do
  for k in lat wide; do
    THOR_HIP_KERNEL=$k timeout 300 python bench.py --width 1920 --height 1080 --streams $s --warmup 5 --steps 4 --no-verify --no-cpu-baseline > $O/r6c13_1080p_s${s}_$k.json 2> $O/r6c13_1080p_s${s}_$k.err
    echo "$(el) 1080p LDB $s streams forced $k: $(line $O/r6c13_1080p_s${s}_$k.json)"
  done
done
