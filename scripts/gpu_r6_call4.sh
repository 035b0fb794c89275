#!/bin/bash
# round 6, call 4: (a) known answers + parity + full-size 10/12-bit and RA goldens on the library with the 16-bit lane-per-candidate full-pel search,
# (b) the cost of the scratch traffic at EQUAL residency (call 3's attempt set THOR_HIP_WGS to an empty string for the spill-free build): the product held to
# 512 resident workgroups measured 128.4 Mpx/s and 10.3 KB/px in call 3; here the -DTK_OCC=2 build (256 VGPRs, no register-pressure spills, 512 workgroups by
# construction) - throughput, FETCH_SIZE / WRITE_SIZE - and the single 3840x2160 stream with it, (c) A/B against the previous commit's library.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py -q -m gpu > $O/r6c4_par.log 2>&1; echo "$(el) parity + kat rc=$? $(tail -1 $O/r6c4_par.log)"; grep -E "^FAILED|^ERROR" $O/r6c4_par.log | head
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "10bit or 12bit or ra_n" > $O/r6c4_big.log 2>&1; echo "$(el) full-size 10/12-bit + RA rc=$? $(tail -1 $O/r6c4_big.log)"; grep -E "^FAILED|^ERROR" $O/r6c4_big.log | head
AB8="--width 1920 --height 1080 --streams 256 --warmup 5 --steps 4 --no-verify --no-cpu-baseline --lockstep"
AB10="--config hdb16 --bitdepth 10 --width 1920 --height 1080 --streams 96 --warmup 1 --steps 16 --no-verify --no-cpu-baseline"
lib=$R/thor_amd/libthor_hip_occ2.so
THOR_HIP_LIB=$lib timeout 300 python bench.py $AB8 > $O/r6c4_ab_occ2.log 2>$O/r6c4_ab_occ2.err
echo "$(el) 1080p s256 P5-P8 lockstep occ2 (512 workgroups, no spills): $(grep -o '"value": [0-9.]*' $O/r6c4_ab_occ2.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r6c4_ab_occ2.log)"; tail -2 $O/r6c4_ab_occ2.err
cd /tmp
for tag in fetch write; do
  c=FETCH_SIZE; [ $tag = write ] && c=WRITE_SIZE
  THOR_HIP_LIB=$lib timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/r6c4pmc_occ2_$tag -- python $R/bench.py $AB8 > $O/r6c4pmc_occ2_$tag.log 2>&1
  echo "$(el) pmc occ2 $tag rc=$? $(grep -o '"value": [0-9.]*' $O/r6c4pmc_occ2_$tag.log | head -1)"
done
(cd $R && python3 scripts/pmc_summary.py gpurun_out/r6c4pmc_occ2 1920 1080 256 4 gpurun_out/r6c4_pmc_occ2 "bench.py $AB8, -DTK_OCC=2 build (256 VGPRs, 512 resident workgroups)" 5 | tail -6)
find $O -name "*_kernel_trace.csv" -size +1M -delete; find $O -name "*_counter_collection.csv" -size +4M -delete; find $O -name "*.csv" -path "*pmc*" -size +1M -delete
cd $R
THOR_HIP_LIB=$lib timeout 400 python bench.py --streams 1 --warmup 5 --steps 20 --verify recorded --no-cpu-baseline > $O/r6c4_s1_occ2.json 2> $O/r6c4_s1_occ2.err
echo "$(el) 4K s1 occ2: $(grep -o '"value": [0-9.]*' $O/r6c4_s1_occ2.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r6c4_s1_occ2.json) $(grep -o '"ms_per_step": [0-9.]*' $O/r6c4_s1_occ2.json)"
for v in pre16 new; do
  lib=$R/thor_amd/libthor_hip_$v.so; [ $v = new ] && lib=$R/thor_amd/libthor_hip.so
  THOR_HIP_LIB=$lib timeout 400 python bench.py $AB10 > $O/r6c4_ab10_$v.log 2>$O/r6c4_ab10_$v.err
  echo "$(el) 1080p 10-bit HDB16 s96 $v: $(grep -o '"value": [0-9.]*' $O/r6c4_ab10_$v.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r6c4_ab10_$v.log) $(grep -o '"filters_reference_creation_bit_gather_kernels": [0-9.]*' $O/r6c4_ab10_$v.log)"
  THOR_HIP_LIB=$lib timeout 400 python bench.py $AB8 > $O/r6c4_ab8_$v.log 2>$O/r6c4_ab8_$v.err
  echo "$(el) 1080p 8-bit LDB s256 $v: $(grep -o '"value": [0-9.]*' $O/r6c4_ab8_$v.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r6c4_ab8_$v.log) $(grep -o '"filters_reference_creation_bit_gather_kernels": [0-9.]*' $O/r6c4_ab8_$v.log)"
done
du -sh $O | tail -1
