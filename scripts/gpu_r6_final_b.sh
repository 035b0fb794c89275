#!/bin/bash
# round 6, final call B: the profiles of THE REGIME THE DRIVER TIMES (python bench.py --warmup 5 --steps 20: 3840x2160 LDB, 128 streams, coded frames 5..24, two
# stream groups half a frame apart), collected on bench.py itself:
#   1. the plain line; 2. rocprofv3 --kernel-trace --stats -> r06_rocprofv3_kernel_stats_bench.md; 3. PMC passes, one counter group per run (--pmc + --kernel-trace
#   only) -> r06_pmc_bench.{md,json}: the 41 launches of the 20 TIMED frames, stamped with the digest of the engine sources; 4. a line that carries roofline.traffic;
#   5. average vector-memory / LDS / scalar-memory instruction latency (SQ_INST_LEVEL_* / SQ_INSTS_*) with 256 and with 768 resident workgroups (1080p).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
timeout 700 python bench.py --steps 20 --warmup 5 > $O/r06_bench_driver_regime.json 2> $O/r6fb_driver.err
echo "$(el) driver regime: $(grep -o '"value": [0-9.]*' $O/r06_bench_driver_regime.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r06_bench_driver_regime.json) $(grep -o '"ms_per_step": [0-9.]*' $O/r06_bench_driver_regime.json | head -1)"
BARGS="--warmup 5 --steps 20 --verify recorded --no-cpu-baseline"
cd /tmp
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r6_rocprof_bench -o bench -- python $R/bench.py $BARGS > $O/r6_rocprof_bench.log 2>&1; echo "$(el) rocprof bench rc=$? $(grep -o '"value": [0-9.]*' $O/r6_rocprof_bench.log | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r6_rocprof_bench.log)"
python3 $R/scripts/kernel_stats_md.py $O/r6_rocprof_bench "rocprofv3 --kernel-trace --stats of the benched workload, the driver's regime (round 6, final library)" "cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py $BARGS (3840x2160 LDB_high_efficiency qp 32, 128 streams in two groups half a frame apart; coded frames 0..24, frames 5..24 timed: 11 + 41 launches of k_superblocks)" > $O/r06_rocprofv3_kernel_stats_bench.md 2>&1; head -14 $O/r06_rocprofv3_kernel_stats_bench.md
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
  timeout 700 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/r6pmc_$tag -- python $R/bench.py $BARGS > $O/r6pmc_$tag.log 2>&1
  echo "$(el) pmc $tag rc=$? $(grep -o '"value": [0-9.]*' $O/r6pmc_$tag.log | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r6pmc_$tag.log)"
}
pmc sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
cd $R
python3 scripts/pmc_summary.py gpurun_out/r6pmc 3840 2160 128 20 gpurun_out/r06_pmc_bench "python bench.py --warmup 5 --steps 20: 3840x2160 LDB_high_efficiency qp 32, 128 closed streams in two groups half a frame apart, the 41 launches of the TIMED coded frames 5..24 (4 references + bi-prediction), final round-6 library" 11 | tail -9
cp gpurun_out/r06_pmc_bench.json gpurun_out/r06_pmc_bench.md profiles/
find $O -name "*_kernel_trace.csv" -size +1M -delete; find $O -name "*_counter_collection.csv" -size +4M -delete
timeout 700 python bench.py --steps 20 --warmup 5 > $O/r06_bench_driver_regime_traffic.json 2> $O/r6fb_driver2.err
echo "$(el) bench with traffic: $(grep -o '"value": [0-9.]*' $O/r06_bench_driver_regime_traffic.json | head -1) $(grep -o '"traffic": [0-9a-z]*' $O/r06_bench_driver_regime_traffic.json) $(grep -o '"bit_exact": [a-z]*' $O/r06_bench_driver_regime_traffic.json)"
AB="--width 1920 --height 1080 --streams 256 --warmup 5 --steps 4 --no-verify --no-cpu-baseline --lockstep"
cd /tmp
for n in 256 768; do
  THOR_HIP_WGS=$n timeout 300 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/r6lat_$n -- python $R/bench.py $AB > $O/r6lat_$n.log 2>&1
  echo "$(el) latency counters, $n resident workgroups rc=$? $(grep -o '"value": [0-9.]*' $O/r6lat_$n.log | head -1)"
  python3 - <<PY
import csv, glob, collections
fs = glob.glob('$O/r6lat_$n/*/*_counter_collection.csv')
if fs:
    a = collections.Counter()
    rows = [r for r in csv.DictReader(open(fs[0])) if 'k_superblocks' in r['Kernel_Name']]
    ids = sorted({int(r['Dispatch_Id']) for r in rows})[5:]
    for r in rows:
        if int(r['Dispatch_Id']) in ids: a[r['Counter_Name']] += float(r['Counter_Value'])
    g = lambda k: a.get(k, float('nan'))
    print('  resident workgroups $n: average latency in cycles: vector memory %.0f, LDS %.0f, scalar memory %.0f' % (g('SQ_INST_LEVEL_VMEM') / (g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR')), g('SQ_INST_LEVEL_LDS') / g('SQ_INSTS_LDS'), g('SQ_INST_LEVEL_SMEM') / g('SQ_INSTS_SMEM')), dict(a))
PY
done
find $O -name "*.csv" -path "*r6lat*" -size +1M -delete
du -sh $O | tail -1
