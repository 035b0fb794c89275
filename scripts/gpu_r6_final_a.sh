#!/bin/bash
# round 6, final call A: the whole GPU suite on the final library, the default bench invocation, and the other BASELINE configurations as verified lines
# with a CPU baseline (same commands as rounds 4 / 5).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
timeout 1800 python -m pytest tests -q -m gpu > $O/r06_gpu_suite_final.log 2>&1; echo "$(el) pytest -m gpu rc=$? $(tail -1 $O/r06_gpu_suite_final.log)"; grep -E "^FAILED|^ERROR" $O/r06_gpu_suite_final.log | head
timeout 900 python bench.py > $O/r06_bench_default.json 2> $O/r6fa_default.err
echo "$(el) default bench: $(grep -o '"value": [0-9.]*' $O/r06_bench_default.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r06_bench_default.json)"
timeout 900 python bench.py --width 1920 --height 1080 --streams 256 --warmup 5 --steps 8 > $O/r06_bench_1080p_ldb.json 2> $O/r6fa_1080p.err
echo "$(el) config 2 (1080p LDB, 256 streams): $(grep -o '"value": [0-9.]*' $O/r06_bench_1080p_ldb.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r06_bench_1080p_ldb.json)"
timeout 900 python bench.py --config ra --streams 96 --warmup 1 --steps 8 --verify recorded --cpu-sample 1920x1080 > $O/r06_bench_4k_ra.json 2> $O/r6fa_ra.err
echo "$(el) config 3 (4K RA qp 27, 96 streams): $(grep -o '"value": [0-9.]*' $O/r06_bench_4k_ra.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r06_bench_4k_ra.json)"
timeout 900 python bench.py --sigma 6 --warmup 5 --steps 2 --verify recorded --cpu-sample 1920x1080 > $O/r06_bench_sigma6.json 2> $O/r6fa_sigma6.err
echo "$(el) hard content (sigma 6, 4K LDB, 128 streams): $(grep -o '"value": [0-9.]*' $O/r06_bench_sigma6.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r06_bench_sigma6.json)"
timeout 900 python bench.py --config hdb16 --bitdepth 10 --streams 96 --warmup 1 --steps 16 --verify recorded --cpu-sample 1920x1080 > $O/r06_bench_4k_hdb16_10bit.json 2> $O/r6fa_cfg5.err
echo "$(el) config 5 (4K 10-bit HDB16, 96 streams): $(grep -o '"value": [0-9.]*' $O/r06_bench_4k_hdb16_10bit.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r06_bench_4k_hdb16_10bit.json)"
du -sh $O | tail -1
