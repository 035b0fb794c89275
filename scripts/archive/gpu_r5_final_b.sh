#!/bin/bash
# round 5, final call B: the profiles of THE REGIME THE DRIVER TIMES (python bench.py --warmup 5 --steps 20: 3840x2160 LDB, 128 streams, coded
# frames 5..24, two stream groups half a frame apart), collected on bench.py itself:
#   1. rocprofv3 --kernel-trace --stats            -> r05_rocprofv3_kernel_stats_bench.md
#   2. PMC passes, one counter group per run (SQ / FETCH_SIZE / WRITE_SIZE; --pmc + --kernel-trace only) -> r05_pmc_bench.{md,json}: the counters of
#      the 41 launches of the 20 TIMED frames (the 11 launches of the warm-up frames are skipped), stamped with the digest of the engine sources
#   3. a bench line that carries roofline.traffic from (2)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
BARGS="--warmup 5 --steps 20 --verify recorded --no-cpu-baseline"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_rocprof_bench -o bench -- python $R/bench.py $BARGS > $O/r5_rocprof_bench.log 2>&1; echo "$(el) rocprof bench rc=$? $(grep -o '"value": [0-9.]*' $O/r5_rocprof_bench.log | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r5_rocprof_bench.log)"
python3 $R/scripts/kernel_stats_md.py $O/r5_rocprof_bench "rocprofv3 --kernel-trace --stats of the benched workload, the driver's regime (round 5, final library)" "cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py $BARGS (3840x2160 LDB_high_efficiency qp 32, 128 streams in two groups half a frame apart; coded frames 0..24, frames 5..24 timed: 11 + 41 launches of k_superblocks)" > $O/r05_rocprofv3_kernel_stats_bench.md 2>&1; head -14 $O/r05_rocprofv3_kernel_stats_bench.md
python3 - <<PY
import csv, glob
f = glob.glob('$O/r5_rocprof_bench/**/*kernel_trace.csv', recursive=True)
rows = sorted((r for r in csv.DictReader(open(f[0])) if 'k_superblocks' in r['Kernel_Name']), key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
print('k_superblocks launches', len(d), 'warm-up (first 11) sum ms %.0f' % sum(d[:11]), 'timed (last 41) sum ms %.0f avg %.1f' % (sum(d[11:]), sum(d[11:]) / max(len(d[11:]), 1)))
print('timed launches ms:', ' '.join('%.0f' % x for x in d[11:]))
PY
rm -rf $O/r5_rocprof_bench
pmc() {
  tag=$1; shift
  timeout 700 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/r5pmc_$tag -- python $R/bench.py $BARGS > $O/r5pmc_$tag.log 2>&1
  echo "$(el) pmc $tag rc=$? $(grep -o '"value": [0-9.]*' $O/r5pmc_$tag.log | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r5pmc_$tag.log)"
}
pmc sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
if [ $(( $(date +%s) - T0 )) -lt 900 ]; then pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES; fi
cd $R
python3 scripts/pmc_summary.py gpurun_out/r5pmc 3840 2160 128 20 gpurun_out/r05_pmc_bench "python bench.py --warmup 5 --steps 20: 3840x2160 LDB_high_efficiency qp 32, 128 closed streams in two groups half a frame apart, the 41 launches of the TIMED coded frames 5..24 (4 references + bi-prediction), final round-5 library" 11 | tail -9
cp gpurun_out/r05_pmc_bench.json gpurun_out/r05_pmc_bench.md profiles/
find $O -name "*_kernel_trace.csv" -path "*r5pmc*" -size +2M -delete
find $O -name "*_counter_collection.csv" -path "*r5pmc*" -size +8M -delete
timeout 700 python bench.py --steps 20 --warmup 5 > $O/r05_bench_driver_regime_traffic.json 2> $O/r05_bench_driver_regime_traffic.err
echo "$(el) bench with traffic: $(grep -o '"value": [0-9.]*' $O/r05_bench_driver_regime_traffic.json | head -1) $(grep -o '"traffic": [0-9a-z]*' $O/r05_bench_driver_regime_traffic.json) $(grep -o '"bit_exact": [a-z]*' $O/r05_bench_driver_regime_traffic.json)"
