#!/bin/bash
# round 4, call 8 (planning data for the next round; profiling variant built from a scratch copy of the sources, patch:
# profiles/r04_prof_intra_phases.diff): where an intra trial spends its time (LDB, 1920x1080 x 128 streams, I + 5 P).
# slots 16..21, 24, 25 of the output = luma edges | luma prediction | luma transform unit | luma bound check (SSD + coefficient bits) |
# chroma part (edges, prediction, CfL, two transform units) | tb-split luma units 0..2 | unsplit intra trials (n) | of them past the luma bound (n)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out /tmp/w
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 7 2
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_pi tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_profin.so -Wl,-rpath,$R/thor_amd
THOR_PROF=1 timeout 200 /tmp/w/thorenc_pi -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 6 -streams 128 -wrap 7 > gpurun_out/r4c8_prof_intra.log 2>&1
echo "rc=$?"; grep -v "^[WIE]2026" gpurun_out/r4c8_prof_intra.log | tail -34
