#!/bin/bash
# round 6, second final call B: the profiles of THE REGIME THE DRIVER TIMES (python bench.py --warmup 5 --steps 20: 3840x2160 LDB, 160 streams - the default since
# call 12 - coded frames 5..24, two stream groups half a frame apart), collected on bench.py itself with the final library:
#   1. rocprofv3 --kernel-trace --stats -> r06_rocprofv3_kernel_stats_bench.md; 2. PMC passes, one counter group per run (--pmc + --kernel-trace only) ->
#   r06_pmc_bench.{md,json}: the 41 launches of the 20 TIMED frames, stamped with the digest of the engine sources; 3. the plain line, which then carries
#   roofline.traffic and the CPU baseline.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
BARGS="--warmup 5 --steps 20 --verify recorded --no-cpu-baseline"
cd /tmp
timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r6_rocprof_bench -o bench -- python $R/bench.py $BARGS > $O/r6_rocprof_bench.log 2>&1; echo "$(el) rocprof bench rc=$? $(grep -o '"value": [0-9.]*' $O/r6_rocprof_bench.log | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r6_rocprof_bench.log)"
python3 $R/scripts/kernel_stats_md.py $O/r6_rocprof_bench "rocprofv3 --kernel-trace --stats of the benched workload, the driver's regime (round 6, final library)" "cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py $BARGS (3840x2160 LDB_high_efficiency qp 32, 160 streams in two groups half a frame apart; coded frames 0..24, frames 5..24 timed: 11 + 41 launches of k_superblocks)" > $O/r06_rocprofv3_kernel_stats_bench.md 2>&1; head -16 $O/r06_rocprofv3_kernel_stats_bench.md
python3 - <<PY
import csv, glob
f = glob.glob('$O/r6_rocprof_bench/**/*kernel_trace.csv', recursive=True)
rows = sorted((r for r in csv.DictReader(open(f[0])) if 'k_superblocks' in r['Kernel_Name']), key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
print('k_superblocks launches', len(d), 'warm-up (first 11) sum ms %.0f' % sum(d[:11]), 'timed (last 41) sum ms %.0f avg %.1f' % (sum(d[11:]), sum(d[11:]) / max(len(d[11:]), 1)))
print('timed launches ms:', ' '.join('%.0f' % x for x in d[11:]))
PY
rm -rf $O/r6_rocprof_bench
pmc() {
  tag=$1; shift
  timeout 800 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/r6pmc_$tag -- python $R/bench.py $BARGS > $O/r6pmc_$tag.log 2>&1
  echo "$(el) pmc $tag rc=$? $(grep -o '"value": [0-9.]*' $O/r6pmc_$tag.log | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r6pmc_$tag.log)"
}
rm -rf $O/r6pmc_sq1 $O/r6pmc_fetch $O/r6pmc_write
pmc sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
cd $R
python3 scripts/pmc_summary.py gpurun_out/r6pmc 3840 2160 160 20 gpurun_out/r06_pmc_bench "python bench.py --warmup 5 --steps 20: 3840x2160 LDB_high_efficiency qp 32, 160 closed streams in two groups half a frame apart, the 41 launches of the TIMED coded frames 5..24 (4 references + bi-prediction), final round-6 library" 11 | tail -9
cp gpurun_out/r06_pmc_bench.json gpurun_out/r06_pmc_bench.md profiles/
find $O -name "*_kernel_trace.csv" -size +1M -delete; find $O -name "*_counter_collection.csv" -size +4M -delete
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06_bench_driver_regime.json 2> $O/r6f2b_driver.err
echo "$(el) driver regime: $(grep -o '"value": [0-9.]*' $O/r06_bench_driver_regime.json | head -1) $(grep -o '"traffic": [0-9a-z]*' $O/r06_bench_driver_regime.json) $(grep -o '"bit_exact": [a-z]*' $O/r06_bench_driver_regime.json) $(grep -o '"ms_per_step": [0-9.]*' $O/r06_bench_driver_regime.json | head -1)"
du -sh $O | tail -1
