#!/bin/bash
# round 2, call 7: validation of the round's final library - full -m gpu suite, smoke, default bench.py under rocprofv3 kernel stats
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r2c7_smoke.log 2>&1; tail -4 gpurun_out/r2c7_smoke.log | head -1
( time timeout 900 python -m pytest tests -m gpu -q --durations=6 ) > gpurun_out/r2c7_tests.log 2>&1
tail -14 gpurun_out/r2c7_tests.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2c7_rocprof -o bench -- python $R/bench.py > $R/gpurun_out/r2c7_bench.log 2>&1; cd $R
grep -v "^[WIE]2026" gpurun_out/r2c7_bench.log | tail -2
find gpurun_out/r2c7_rocprof -name "*.csv" | head
