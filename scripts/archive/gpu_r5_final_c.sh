#!/bin/bash
# round 5, final call C: the other BASELINE configurations with the final library, verified inside the run, with a CPU baseline (same commands as
# round 4's lines, profiles/r04_bench_*.json), and one extra operating point of the headline workload (192 streams).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
line() { echo "$(el) $1: $(grep -o '"value": [0-9.]*' $2 | head -1) $(grep -o '"ms_per_step": [0-9.]*' $2) $(grep -o '"bit_exact": [a-z]*' $2) $(grep -o '"avg_launch_ms": [0-9.]*' $2) $(grep -o '"cpu_baseline": {"value": [0-9.a-z]*' $2)"; }
timeout 400 python bench.py --width 1920 --height 1080 --streams 256 --warmup 5 --steps 8 > $O/r05_bench_1080p_ldb.json 2> $O/r05_bench_1080p_ldb.err; line "cfg 2: 1080p LDB s256" $O/r05_bench_1080p_ldb.json; tail -2 $O/r05_bench_1080p_ldb.err
timeout 400 python bench.py --config ra --streams 96 --warmup 1 --steps 8 --verify recorded --cpu-sample 1920x1080 > $O/r05_bench_4k_ra.json 2> $O/r05_bench_4k_ra.err; line "cfg 3: 4K RA s96" $O/r05_bench_4k_ra.json; tail -2 $O/r05_bench_4k_ra.err
timeout 500 python bench.py --config hdb16 --bitdepth 10 --streams 96 --warmup 1 --steps 16 --verify recorded --cpu-sample 1920x1080 > $O/r05_bench_4k_hdb16_10bit.json 2> $O/r05_bench_4k_hdb16_10bit.err; line "cfg 5: 4K 10-bit HDB16 s96" $O/r05_bench_4k_hdb16_10bit.json; tail -2 $O/r05_bench_4k_hdb16_10bit.err
timeout 300 python bench.py --sigma 6 --warmup 5 --steps 2 --verify recorded --cpu-sample 1920x1080 > $O/r05_bench_sigma6.json 2> $O/r05_bench_sigma6.err; line "hard content (sigma 6)" $O/r05_bench_sigma6.json; tail -2 $O/r05_bench_sigma6.err
# what FETCH_SIZE / WRITE_SIZE count for a small working set that is re-written and re-read in place (tools/ubench_rewrite.cpp)
cd /tmp
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/r5cal_$c -- $R/tools/ubench_rewrite > $O/r5cal_$c.log 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob('$O/r5cal_$c/**/*_counter_collection.csv', recursive=True):
    for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r['Dispatch_Id'])):
        print('rewrite', r['Kernel_Name'][:28], r['Counter_Name'], r['Counter_Value'], 'KiB')
PY
done
grep "^k_" $O/r5cal_WRITE_SIZE.log
cd $R
if [ $(( $(date +%s) - T0 )) -lt 620 ]; then
  timeout 500 python bench.py --streams 192 --warmup 5 --steps 20 --verify recorded --no-cpu-baseline > $O/r05_bench_4k_ldb_s192.json 2> $O/r05_bench_4k_ldb_s192.err; line "headline workload with 192 streams" $O/r05_bench_4k_ldb_s192.json; tail -2 $O/r05_bench_4k_ldb_s192.err
fi
