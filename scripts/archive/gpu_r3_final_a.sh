#!/bin/bash
# round 3, final A: the complete -m gpu suite on the final library (thor_amd/libthor_hip.so built by __graft_entry__.build())
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r3_gpu_suite_final.log 2>&1; echo "suite rc=$?"; tail -22 gpurun_out/r3_gpu_suite_final.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke.log 2>&1; echo "smoke rc=$? $(tail -1 gpurun_out/r3_smoke.log)"
