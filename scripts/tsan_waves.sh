#!/bin/bash
# Data-race check of the wave-parallel block decision: 4-wave host simulation (one OS thread per wavefront of a workgroup:
# work queue, shared minimum key, lock-step bi-prediction, snapshots) under ThreadSanitizer.
# Expected reports: only the host-only debug counters g_prune_stat (tk_block.h, #if TK_HOST).
set -e
cd "$(dirname "$0")/.."
g++ -std=c++17 -O1 -g -fno-strict-aliasing -fsanitize=thread -DTHOR_HOSTSIM -DTHOR_HOSTSIM_WAVES=4 -ffp-contract=off -pthread -o /tmp/hostsim_tsan_w4 tests/hostsim/hostsim.cpp
python3 -m thor_amd.synth /tmp/tsan_clip.yuv 192 128 3 9
TSAN_OPTIONS="halt_on_error=0" /tmp/hostsim_tsan_w4 -cf configs/ldb_high_efficiency.cfg -if /tmp/tsan_clip.yuv -width 192 -height 128 -qp 36 -n 3 -f 30 -of /tmp/tsan.bit -rf /tmp/tsan.yuv 2>&1 | grep -A4 "WARNING: ThreadSanitizer" | grep "#0" | sort | uniq -c
