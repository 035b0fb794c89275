#!/bin/bash
# round 4, call 13 (planning data; -DTHOR_PROF build of the final sources): phase profile over 14 frames (I + 13 P, crossing the high-quality
# frame 12) next to the 6-frame profile of call 7 - what the later, harder frames of the driver's regime are made of.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out /tmp/w
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 15 2
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_prof tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_prof.so -Wl,-rpath,$R/thor_amd
THOR_PROF=1 timeout 300 /tmp/w/thorenc_prof -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 14 -streams 128 -wrap 15 > gpurun_out/r4c13_prof_ldb_n14.log 2>&1
echo "rc=$?"; grep -v "^[WIE]2026" gpurun_out/r4c13_prof_ldb_n14.log | tail -34
