#!/bin/bash
# round 6, call 3: (a) the new known-answer families + parity suite on the product library, (b) what the scratch traffic costs at EQUAL residency: the
# product (168 VGPRs, spills) held to 512 resident workgroups against the -DTK_OCC=2 build (256 VGPRs: no register-pressure spills; 512 workgroups by
# construction) - throughput and FETCH_SIZE / WRITE_SIZE, (c) the single-stream point with the spill-free build, (d) the default bench line with the
# round-6 measurement fields (same-box full-geometry cpu_baseline, measured H2D), (e) the 12-bit 3840x2160 golden.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
timeout 900 python -m pytest tests/test_gpu_kat.py tests/test_gpu_parity.py -q -m gpu > $O/r6c3_par.log 2>&1; echo "$(el) kat + parity rc=$? $(tail -1 $O/r6c3_par.log)"; grep -E "^FAILED|^ERROR" $O/r6c3_par.log | head -20
AB="--width 1920 --height 1080 --streams 256 --warmup 5 --steps 4 --no-verify --no-cpu-baseline --lockstep"
cd /tmp
for v in new512 occ2; do
  lib=$R/thor_amd/libthor_hip_occ2.so; wgs=""; [ $v = new512 ] && { lib=$R/thor_amd/libthor_hip.so; wgs=512; }
  THOR_HIP_WGS=$wgs THOR_HIP_LIB=$lib timeout 300 python $R/bench.py $AB > $O/r6c3_ab_$v.log 2>$O/r6c3_ab_$v.err
  echo "$(el) 1080p s256 P5-P8 lockstep $v: $(grep -o '"value": [0-9.]*' $O/r6c3_ab_$v.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r6c3_ab_$v.log)"
  pmc() { tag=$1; shift
    THOR_HIP_WGS=$wgs THOR_HIP_LIB=$lib timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/r6c3pmc_${v}_$tag -- python $R/bench.py $AB > $O/r6c3pmc_${v}_$tag.log 2>&1
    echo "$(el) pmc $v $tag rc=$? $(grep -o '"value": [0-9.]*' $O/r6c3pmc_${v}_$tag.log | head -1)"; }
  pmc fetch FETCH_SIZE
  pmc write WRITE_SIZE
  pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY
  (cd $R && python3 scripts/pmc_summary.py gpurun_out/r6c3pmc_$v 1920 1080 256 4 gpurun_out/r6c3_pmc_$v "bench.py $AB, library variant $v (512 resident workgroups)" 5 | tail -6)
done
find $O -name "*_kernel_trace.csv" -path "*r6c3pmc*" -size +2M -delete; find $O -name "*_counter_collection.csv" -path "*r6c3pmc*" -size +8M -delete
cd $R
THOR_HIP_LIB=$R/thor_amd/libthor_hip_occ2.so timeout 600 python bench.py --streams 1 --warmup 5 --steps 20 --verify recorded --no-cpu-baseline > $O/r6c3_s1_occ2.json 2> $O/r6c3_s1_occ2.err
echo "$(el) 4K s1 occ2: $(grep -o '"value": [0-9.]*' $O/r6c3_s1_occ2.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r6c3_s1_occ2.json) $(grep -o '"ms_per_step": [0-9.]*' $O/r6c3_s1_occ2.json)"
timeout 900 python bench.py > $O/r6c3_bench_default.json 2> $O/r6c3_bench_default.err
echo "$(el) default bench: $(grep -o '"value": [0-9.]*' $O/r6c3_bench_default.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r6c3_bench_default.json)"; python3 -c "
import json; d=json.loads(open('$O/r6c3_bench_default.json').read().strip().splitlines()[-1]); print(json.dumps(d['cpu_baseline'])[:1500]); print(json.dumps(d['io']['h2d']))"
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "12bit" > $O/r6c3_big12.log 2>&1; echo "$(el) 4K 12-bit golden rc=$? $(tail -1 $O/r6c3_big12.log)"
