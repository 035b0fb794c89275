#!/bin/bash
# Data-race check of the lane-parallel engine logic: 4-lane host simulation (one OS thread per lane) under ThreadSanitizer.
# Expected reports: only same-value stores by all lanes to team-shared state (make_ws pointers, nd.syn.mvp) and the
# host-only debug counters g_prune_stat; anything else is a missing t.sync().
set -e
cd "$(dirname "$0")/.."
g++ -std=c++17 -O1 -g -fno-strict-aliasing -fsanitize=thread -DTHOR_HOSTSIM -DTHOR_HOSTSIM_LANES=4 -ffp-contract=off -pthread -o /tmp/hostsim_tsan tests/hostsim/hostsim.cpp
python3 -m thor_amd.synth /tmp/tsan_clip.yuv 192 128 2 9
TSAN_OPTIONS="halt_on_error=0" /tmp/hostsim_tsan -cf configs/ldb_high_efficiency.cfg -if /tmp/tsan_clip.yuv -width 192 -height 128 -qp 36 -n 2 -f 30 -of /tmp/tsan.bit -rf /tmp/tsan.yuv 2>&1 | grep -A3 "WARNING: ThreadSanitizer" | grep "#0" | sort | uniq -c
