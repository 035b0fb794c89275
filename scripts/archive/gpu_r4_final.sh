#!/bin/bash
# round 4, final call: everything that is reported about the library that ships.
#   1. full -m gpu suite
#   2. PMC passes (SQ / FETCH_SIZE / WRITE_SIZE) of the benched geometry through tools/thorenc_hip -> r04_pmc_bench.{md,json} (stamped with
#      the digest of the engine sources; bench.py attaches the figures to lines of the same sources only), WRITE_SIZE / FETCH_SIZE
#      calibration of the store patterns (tools/ubench_write)
#   3. the driver's regime: python bench.py --steps 20 --warmup 5 (self-verifying, cpu_baseline)
#   4. rocprofv3 --kernel-trace --stats of the same command (short)
#   5. BASELINE configs 2 / 3 / 5: verified lines with a CPU baseline (1080p LDB live; 4K RA and 4K 10-bit HDB16 against recorded
#      reference runs, CPU sample on a 1080p crop)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
O=$R/gpurun_out
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s]"; }
line() { echo "$(el) $1: $(grep -o '"value": [0-9.]*' $2 | head -1) $(grep -o '"ms_per_step": [0-9.]*' $2) $(grep -o '"bit_exact": [a-z]*' $2) $(grep -o '"cpu_baseline": {"value": [0-9.a-z]*' $2)"; }
timeout 1000 python -m pytest tests -q -x -m gpu --durations=6 > $O/r4f_suite.log 2>&1; echo "$(el) full -m gpu suite rc=$? $(tail -1 $O/r4f_suite.log)"
# ---- 2. PMC
python3 -m thor_amd.synth /tmp/w/uhd.yuv 3840 2160 7 4
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc $R/tools/thorenc_hip.c -L$R/thor_amd -lthor_hip -Wl,-rpath,$R/thor_amd
PARGS="-cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/uhd.yuv -width 3840 -height 2160 -qp 32 -f 30 -n 6 -streams 128 -wrap 7"
cd /tmp
pmc() {
  tag=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/r4pmc_$tag -- /tmp/w/thorenc $PARGS > $O/r4pmc_$tag.log 2>&1
  echo "$(el) pmc $tag rc=$? $(grep thorenc_hip: $O/r4pmc_$tag.log | cut -c1-160)"
}
pmc sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS
pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
cd $R
python3 scripts/pmc_summary.py gpurun_out/r4pmc 3840 2160 128 6 gpurun_out/r04_pmc_bench "3840x2160 LDB_high_efficiency qp 32, 128 closed streams x (I + 5 P; P4 and P5 search 4 references) through tools/thorenc_hip, final round-4 library" | tail -9
cp gpurun_out/r04_pmc_bench.json gpurun_out/r04_pmc_bench.md profiles/   # the bench runs below attach the traffic of these very sources
cd /tmp
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/r4cal_$c -- $R/tools/ubench_write > $O/r4cal_$c.log 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob('$O/r4cal_$c/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        print('calibration', r['Kernel_Name'][:24], r['Counter_Name'], r['Counter_Value'])
PY
done
grep -v "^[WIE]2026" $O/r4cal_WRITE_SIZE.log | grep "^k_" 
find $O -name "*_kernel_trace.csv" -path "*r4pmc*" -size +2M -delete
cd $R
# ---- 3. driver regime
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r04_bench_driver_regime.json 2> $O/r04_bench_driver_regime.err; line "driver regime" $O/r04_bench_driver_regime.json; tail -2 $O/r04_bench_driver_regime.err
# ---- 4. kernel statistics
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_rocprof_bench -o bench -- python $R/bench.py --warmup 5 --steps 2 --no-verify --no-cpu-baseline > $O/r4_rocprof_bench.log 2>&1; echo "$(el) rocprof bench rc=$?"
python3 $R/scripts/kernel_stats_md.py $O/r4_rocprof_bench "rocprofv3 --kernel-trace --stats of the benched workload (round 4, final library)" "cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --warmup 5 --steps 2 --no-verify --no-cpu-baseline (3840x2160 LDB_high_efficiency qp 32, 128 streams; frames I, P1..P6, the last two timed)" > $O/r04_rocprofv3_kernel_stats_bench.md 2>&1; head -12 $O/r04_rocprofv3_kernel_stats_bench.md
rm -rf $O/r4_rocprof_bench
cd $R
# ---- 5. the other BASELINE configurations, verified
timeout 500 python bench.py --width 1920 --height 1080 --streams 256 --warmup 5 --steps 8 > $O/r04_bench_1080p_ldb.json 2> $O/r04_bench_1080p_ldb.err; line "cfg 2: 1080p LDB s256" $O/r04_bench_1080p_ldb.json
timeout 600 python bench.py --config ra --streams 96 --warmup 1 --steps 8 --verify recorded --cpu-sample 1920x1080 > $O/r04_bench_4k_ra.json 2> $O/r04_bench_4k_ra.err; line "cfg 3: 4K RA s96" $O/r04_bench_4k_ra.json; tail -2 $O/r04_bench_4k_ra.err
timeout 800 python bench.py --config hdb16 --bitdepth 10 --streams 96 --warmup 1 --steps 16 --verify recorded --cpu-sample 1920x1080 > $O/r04_bench_4k_hdb16_10bit.json 2> $O/r04_bench_4k_hdb16_10bit.err; line "cfg 5: 4K 10-bit HDB16 s96" $O/r04_bench_4k_hdb16_10bit.json; tail -2 $O/r04_bench_4k_hdb16_10bit.err
timeout 500 python bench.py --sigma 6 --warmup 5 --steps 2 --verify recorded --cpu-sample 1920x1080 > $O/r04_bench_sigma6.json 2> $O/r04_bench_sigma6.err; line "hard content (sigma 6)" $O/r04_bench_sigma6.json
# ---- 6. bisection of the instrumentation hang (THOR_PROF_MD build, call 2): which counters make k_superblocks hang, and does a wait trip the trap?
#      libthor_hip_profmd (all MD counters), _md1 (work-queue items inside md_worker_sp's loop), _md2 (the trial items' wait), _md4 (master
#      phases), _md9 (= md1 with register accumulators, one store after the loop).  rc 124 = killed by timeout (hang), 134 = aborted (trap / scheduler error).
python3 -m thor_amd.synth /tmp/w/sd.yuv 640 384 5 2
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 5 2
hang() {  # tag lib clip w h streams
  gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_$1 tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_$2.so -Wl,-rpath,$R/thor_amd
  THOR_PROF=md THOR_HIP_SPIN_TIMEOUT_S=20 timeout 50 /tmp/w/thorenc_$1 -cf $R/configs/ldb_high_efficiency.cfg -if $3 -width $4 -height $5 -qp 32 -f 30 -n 4 -streams $6 -wrap 5 > $O/r4f_hang_$1.log 2>&1
  rc=$?; echo "$(el) hang test $1: rc=$rc $(grep -E 'thorenc_hip:|aborted|scheduler failed' $O/r4f_hang_$1.log | cut -c1-150)"; return $rc
}
if [ $(( $(date +%s) - T0 )) -lt 2300 ]; then
  hang profmd_small profmd /tmp/w/sd.yuv 640 384 24
  if [ $? -ne 0 ]; then
    for v in md1 md2 md4 md9; do hang ${v}_small $v /tmp/w/sd.yuv 640 384 24; done
  else
    hang profmd_hd profmd /tmp/w/hd.yuv 1920 1080 64
    if [ $? -ne 0 ]; then for v in md1 md2 md4 md9; do hang ${v}_hd $v /tmp/w/hd.yuv 1920 1080 64; done; fi
  fi
fi
