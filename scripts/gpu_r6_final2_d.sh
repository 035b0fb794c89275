#!/bin/bash
# round 6, second final call D: the library as it ships (rebuilt after the comment-only edits that followed call C): smoke(), the parity tests of the three
# kernels on the small goldens, and a short verified bench line.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; echo "$(el) smoke rc=$?"
timeout 200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "192x128_n3_q32 or kernel_keeps or two_streams" 2>&1 | tail -1; echo "$(el) parity subset"
timeout 200 python bench.py --width 1920 --height 1080 --streams 16 --warmup 1 --steps 3 --no-cpu-baseline > $O/r6f2d_bench.json 2> $O/r6f2d_bench.err
echo "$(el) 1080p x 16 streams, P1..P3 verified live: $(grep -o '"value": [0-9.]*' $O/r6f2d_bench.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r6f2d_bench.json) $(grep -o '"superblock_kernel": {[^}]*}' $O/r6f2d_bench.json | cut -c1-90)"
