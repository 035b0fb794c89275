#!/bin/bash
# round 6, second final call C: the selection rule of call 13 (wide build up to 2.5 x CUs of superblocks in flight) - the whole GPU suite again on the library
# that ships, and a verified mid-stream-count line that now runs on the wide build (1920x1080 x 64 streams, live reference runs).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
timeout 1500 python -m pytest tests -q -m gpu > $O/r06_gpu_suite_final.log 2>&1; echo "$(el) pytest -m gpu rc=$? $(tail -1 $O/r06_gpu_suite_final.log)"; grep -E "^FAILED|^ERROR" $O/r06_gpu_suite_final.log | head
timeout 600 python bench.py --width 1920 --height 1080 --streams 64 --warmup 5 --steps 4 --no-cpu-baseline > $O/r06_bench_1080p_ldb_s64.json 2> $O/r6f2c_s64.err
echo "$(el) 1080p LDB 64 streams (verified live): $(grep -o '"value": [0-9.]*' $O/r06_bench_1080p_ldb_s64.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r06_bench_1080p_ldb_s64.json) $(grep -o '"superblock_kernel": {[^}]*}' $O/r06_bench_1080p_ldb_s64.json | cut -c1-100)"
