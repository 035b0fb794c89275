#!/bin/bash
# round 6, call 8: A/B of -DTK_PIY_INLINE (pred_inter_yuv inlined into its callers: no callee-saved saves per inter trial) against the product: throughput,
# traffic, vector-memory instruction counts and in-flight time (1080p x 256 streams, coded frames 5..8), occupancy of the variant.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
AB="--width 1920 --height 1080 --streams 256 --warmup 5 --steps 4 --no-verify --no-cpu-baseline --lockstep"
for v in new piy new piy; do
  lib=$R/thor_amd/libthor_hip_$v.so; [ $v = new ] && lib=$R/thor_amd/libthor_hip.so
  THOR_HIP_LIB=$lib timeout 300 python bench.py $AB > $O/r6c8_ab_$v.log 2>$O/r6c8_ab_$v.err
  echo "$(el) 1080p s256 P5-P8 lockstep $v: $(grep -o '"value": [0-9.]*' $O/r6c8_ab_$v.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r6c8_ab_$v.log) $(grep -o '"superblock_kernel": {[^}]*}' $O/r6c8_ab_$v.log)"
done
cd /tmp
lib=$R/thor_amd/libthor_hip_piy.so
pmc() { tag=$1; shift
  THOR_HIP_LIB=$lib timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/r6c8pmc_piy_$tag -- python $R/bench.py $AB > $O/r6c8pmc_piy_$tag.log 2>&1
  echo "$(el) pmc piy $tag rc=$? $(grep -o '"value": [0-9.]*' $O/r6c8pmc_piy_$tag.log | head -1)"; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY
(cd $R && python3 scripts/pmc_summary.py gpurun_out/r6c8pmc_piy 1920 1080 256 4 gpurun_out/r6c8_pmc_piy "bench.py $AB, -DTK_PIY_INLINE" 5 | tail -6)
find $O -name "*.csv" -path "*pmc*" -size +1M -delete
