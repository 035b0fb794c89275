#!/bin/bash
# round 6, call 12: the third build of the 8-bit kernel in the product (thor_hip_wide.cpp: eight wavefronts per workgroup, chosen when the streams of a run
# never offer more superblocks than there are CUs): parity of every 8-bit golden through std / lat / wide, the few-stream lines (1 and 8 streams, verified),
# the selection threshold (16 streams: lat against wide), and the driver's regime with 144 / 160 streams (call 11: the default invocation gains 3-5 %).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
line() { echo "$(grep -o '"value": [0-9.]*' $1 | head -1) $(grep -o '"bit_exact": [a-z]*' $1) $(grep -o '"superblock_kernel": {[^}]*}' $1 | cut -c1-120)"; }
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/r6c12_parity.log 2>&1; echo "$(el) parity rc=$? $(tail -1 $O/r6c12_parity.log)"; grep -E "^FAILED|^ERROR" $O/r6c12_parity.log | head
for s in 1 8; do
  timeout 900 python bench.py --streams $s --warmup 5 --steps 20 --no-cpu-baseline > $O/r6c12_s${s}.json 2> $O/r6c12_s${s}.err
  echo "$(el) 4K LDB $s stream(s), product (verified): $(line $O/r6c12_s${s}.json)"; tail -1 $O/r6c12_s${s}.err | cut -c1-200
done
for k in lat wide; do
  THOR_HIP_KERNEL=$k timeout 600 python bench.py --streams 8 --warmup 5 --steps 8 --no-verify --no-cpu-baseline > $O/r6c12_s8_$k.json 2> $O/r6c12_s8_$k.err
  echo "$(el) 4K LDB 8 streams forced $k: $(line $O/r6c12_s8_$k.json)"
  THOR_HIP_KERNEL=$k timeout 600 python bench.py --streams 16 --warmup 5 --steps 8 --no-verify --no-cpu-baseline > $O/r6c12_s16_$k.json 2> $O/r6c12_s16_$k.err
  echo "$(el) 4K LDB 16 streams forced $k: $(line $O/r6c12_s16_$k.json)"
done
for s in 160 144; do
  timeout 900 python bench.py --streams $s --warmup 5 --steps 20 --no-verify --no-cpu-baseline > $O/r6c12_driver_s$s.json 2> $O/r6c12_driver_s$s.err
  echo "$(el) driver's regime with $s streams (not verified): $(line $O/r6c12_driver_s$s.json | cut -c1-40) $(grep -o '"ms_per_step": [0-9.]*' $O/r6c12_driver_s$s.json)"
done
