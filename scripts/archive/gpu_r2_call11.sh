#!/bin/bash
# round 2, call 11: phase profile (THOR_PROF build) of the 4-reference regime: 1080p, 128 streams, I + 5 P
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
python3 -m thor_amd.synth /tmp/w/cif.yuv 416 240 4 7
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/t_pv6 tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_pv6.so -Wl,-rpath,$R/thor_amd
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/t_pv3 tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_pv3.so -Wl,-rpath,$R/thor_amd
( THOR_PROF=1 THOR_HIP_SPIN_TIMEOUT_S=10 timeout 25 stdbuf -o0 -e0 /tmp/w/t_pv6 -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/cif.yuv -width 416 -height 240 -qp 32 -f 30 -n 2 -streams 1 ) > gpurun_out/r2c11_pv6_small.log 2>&1
rc=$?; echo "pv6 small rc=$rc"
T=/tmp/w/t_pv6; [ $rc -ne 0 ] && T=/tmp/w/t_pv3
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 9 2
( time THOR_PROF=1 timeout 400 stdbuf -o0 -e0 $T -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 6 -streams 128 -wrap 7 ) > gpurun_out/r2c11_prof_1080p_s128.log 2>&1
echo "used $T"; cat gpurun_out/r2c11_prof_1080p_s128.log | head -40
