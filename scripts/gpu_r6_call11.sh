#!/bin/bash
# round 6, call 11: (a) workgroups of EIGHT wavefronts for the few-stream operating point (thor_amd/libthor_hip_w8.so = the same sources with
# -DTK_WAVES=8 -DTK_OCC=2: 256 VGPRs, 150 KB of LDS, one workgroup per CU) against the product's two kernels on one and on eight 3840x2160 streams;
# (b) s_setprio of the master wavefront (libthor_hip_mp.so, THOR_HIP_MASTER_PRIO); (c) 144 / 160 streams on the default invocation.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
line() { echo "$(grep -o '"value": [0-9.]*' $1 | head -1) $(grep -o '"bit_exact": [a-z]*' $1) $(grep -o '"superblock_kernel": {[^}]*}' $1)"; }
for s in 1 8; do
  THOR_HIP_LIB=$R/thor_amd/libthor_hip_w8.so THOR_HIP_KERNEL=std timeout 500 python bench.py --streams $s --warmup 5 --steps 20 --verify recorded --no-cpu-baseline > $O/r6c11_s${s}_w8.json 2> $O/r6c11_s${s}_w8.err
  echo "$(el) 4K LDB $s stream(s), 8-wave workgroups: $(line $O/r6c11_s${s}_w8.json)"; tail -2 $O/r6c11_s${s}_w8.err | cut -c1-300
  timeout 500 python bench.py --streams $s --warmup 5 --steps 20 --verify recorded --no-cpu-baseline > $O/r6c11_s${s}_prod.json 2> $O/r6c11_s${s}_prod.err
  echo "$(el) 4K LDB $s stream(s), product: $(line $O/r6c11_s${s}_prod.json)"
done
AB="--width 1920 --height 1080 --streams 256 --warmup 5 --steps 4 --no-cpu-baseline"
for pr in 0 2 3 0; do
  THOR_HIP_LIB=$R/thor_amd/libthor_hip_mp.so THOR_HIP_MASTER=wave0 THOR_HIP_MASTER_PRIO=$pr timeout 400 python bench.py $AB > $O/r6c11_prio$pr.log 2>$O/r6c11_prio$pr.err
  echo "$(el) 1080p s256 master prio $pr: $(line $O/r6c11_prio$pr.log | cut -c1-60)"
done
for s in 144 160; do
  timeout 600 python bench.py --streams $s --no-cpu-baseline > $O/r6c11_default_s$s.json 2> $O/r6c11_default_s$s.err
  echo "$(el) default invocation with $s streams: $(line $O/r6c11_default_s$s.json | cut -c1-60)"
done
