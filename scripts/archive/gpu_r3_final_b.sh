#!/bin/bash
# round 3, final B: the benched workload (3840x2160 LDB_high_efficiency, 128 streams, driver regime = 4-reference frames):
# self-verifying bench line, rocprofv3 kernel statistics of the same command, PMC passes (SQ / FETCH_SIZE / WRITE_SIZE) of the
# same geometry through tools/thorenc_hip, FETCH_SIZE calibration micro-benchmark
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
O=$R/gpurun_out
timeout 900 python bench.py --warmup 5 --steps 2 > $O/r3_bench_driver_regime.json 2> $O/r3_bench_driver_regime.err; echo "bench rc=$?"; cut -c1-1500 $O/r3_bench_driver_regime.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_rocprof_bench -o bench -- python $R/bench.py --warmup 5 --steps 1 --no-verify --no-cpu-baseline > $O/r3_rocprof_bench.log 2>&1; echo "rocprof bench rc=$?"
python3 $R/scripts/kernel_stats_md.py $O/r3_rocprof_bench "rocprofv3 --kernel-trace --stats of the benched workload (round 3, final library)" "cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --warmup 5 --steps 1 --no-verify --no-cpu-baseline (3840x2160 LDB_high_efficiency qp 32, 128 streams; frames I, P1..P5, the last one timed)" > $O/r3_rocprofv3_kernel_stats_bench.md 2>&1; head -12 $O/r3_rocprofv3_kernel_stats_bench.md
rm -rf $O/r3_rocprof_bench
python3 -m thor_amd.synth /tmp/w/uhd.yuv 3840 2160 7 4
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc $R/tools/thorenc_hip.c -L$R/thor_amd -lthor_hip -Wl,-rpath,$R/thor_amd
PARGS="-cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/uhd.yuv -width 3840 -height 2160 -qp 32 -f 30 -n 6 -streams 128 -wrap 7"
pmc() {
  tag=$1; shift
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/r3pmc_$tag -- /tmp/w/thorenc $PARGS > $O/r3pmc_$tag.log 2>&1
  echo "pmc $tag rc=$? $(grep thorenc_hip: $O/r3pmc_$tag.log | cut -c1-160)"
}
pmc sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
cd $R
python3 scripts/pmc_summary.py gpurun_out/r3pmc 3840 2160 128 6 gpurun_out/r03_pmc_bench "3840x2160 LDB_high_efficiency qp 32, 128 closed streams x (I + 5 P; P4 and P5 search 4 references) through tools/thorenc_hip, final round-3 library" | tail -9
cd /tmp
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r3pmc_cal -- $R/tools/ubench_fetch > $O/r3_fetch_calibration.log 2>&1; echo "cal rc=$?"
python3 - <<PY
import csv, glob
for f in glob.glob('$O/r3pmc_cal/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        print('calibration', r['Kernel_Name'][:30], r['Counter_Name'], r['Counter_Value'])
PY
grep -v "^[WIE]2026" $O/r3_fetch_calibration.log | tail -4
find $O -name "*_kernel_trace.csv" -path "*r3pmc*" -size +2M -delete
