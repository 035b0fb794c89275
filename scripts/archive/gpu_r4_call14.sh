#!/bin/bash
# round 4, call 14 (planning data; -DTHOR_PROF -DTHOR_PROF_ME build of the final sources): motion-search cycles and calls by coding-block
# size over 14 frames (I + 13 P) - which block sizes the searches of the later frames belong to.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out /tmp/w
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 15 2
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_pm tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_profme.so -Wl,-rpath,$R/thor_amd
THOR_PROF=me timeout 300 /tmp/w/thorenc_pm -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 14 -streams 128 -wrap 15 > gpurun_out/r4c14_prof_me_n14.log 2>&1
echo "rc=$?"; grep -v "^[WIE]2026" gpurun_out/r4c14_prof_me_n14.log | grep -E "thorenc_hip|sb_total|me_|me cb"
