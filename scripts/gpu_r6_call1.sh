#!/bin/bash
# round 6, call 1: (a) parity of the library with the fork/join functions of a block decision inlined into the kernel (+ -fno-strict-aliasing),
# (b) A/B against the same sources with the calls (-DTK_MDW_CALL = round 5's code), throughput and HBM-side traffic (FETCH_SIZE / WRITE_SIZE /
# vector-memory instruction counts, one --pmc pass each), (c) the single-stream and 8-stream operating points at 3840x2160.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py -q -x -m gpu > $O/r6c1_par.log 2>&1; echo "$(el) parity rc=$? $(tail -1 $O/r6c1_par.log)"
AB="--width 1920 --height 1080 --streams 256 --warmup 5 --steps 4 --no-verify --no-cpu-baseline --lockstep"
for v in call new; do
  lib=$R/thor_amd/libthor_hip_$v.so; [ $v = new ] && lib=$R/thor_amd/libthor_hip.so
  THOR_HIP_LIB=$lib timeout 300 python bench.py $AB > $O/r6c1_ab_$v.log 2>$O/r6c1_ab_$v.err
  echo "$(el) 1080p s256 P5-P8 lockstep $v: $(grep -o '"value": [0-9.]*' $O/r6c1_ab_$v.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r6c1_ab_$v.log)"
done
cd /tmp
for v in call new; do
  lib=$R/thor_amd/libthor_hip_$v.so; [ $v = new ] && lib=$R/thor_amd/libthor_hip.so
  pmc() { tag=$1; shift
    THOR_HIP_LIB=$lib timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/r6c1pmc_${v}_$tag -- python $R/bench.py $AB > $O/r6c1pmc_${v}_$tag.log 2>&1
    echo "$(el) pmc $v $tag rc=$? $(grep -o '"value": [0-9.]*' $O/r6c1pmc_${v}_$tag.log | head -1)"; }
  pmc fetch FETCH_SIZE
  pmc write WRITE_SIZE
  pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY
  (cd $R && python3 scripts/pmc_summary.py gpurun_out/r6c1pmc_$v 1920 1080 256 4 gpurun_out/r6c1_pmc_$v "bench.py $AB, library variant $v" 5 | tail -6)
done
find $O -name "*_kernel_trace.csv" -path "*r6c1pmc*" -size +2M -delete; find $O -name "*_counter_collection.csv" -path "*r6c1pmc*" -size +8M -delete
cd $R
timeout 600 python bench.py --streams 1 --warmup 5 --steps 20 --verify recorded --no-cpu-baseline > $O/r6c1_s1.json 2> $O/r6c1_s1.err
echo "$(el) 4K s1: $(grep -o '"value": [0-9.]*' $O/r6c1_s1.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r6c1_s1.json) $(grep -o '"ms_per_step": [0-9.]*' $O/r6c1_s1.json)"
timeout 600 python bench.py --streams 8 --warmup 5 --steps 20 --no-verify --no-cpu-baseline > $O/r6c1_s8.json 2> $O/r6c1_s8.err
echo "$(el) 4K s8: $(grep -o '"value": [0-9.]*' $O/r6c1_s8.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r6c1_s8.json)"
THOR_HIP_LIB=$R/thor_amd/libthor_hip_call.so timeout 600 python bench.py --streams 1 --warmup 5 --steps 20 --verify recorded --no-cpu-baseline > $O/r6c1_s1_call.json 2> $O/r6c1_s1_call.err
echo "$(el) 4K s1 (call variant): $(grep -o '"value": [0-9.]*' $O/r6c1_s1_call.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r6c1_s1_call.json)"
