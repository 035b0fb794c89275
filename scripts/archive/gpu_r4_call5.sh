#!/bin/bash
# round 4, call 5 (short, after the final call; the product library is untouched): two more bisection builds of the instrumentation hang
# (profiles/r04_hang_bisect_extra.diff: THOR_PROF_MD_PARTS 17 = the loop counters through an LDS-typed pointer, 33 = updated by every lane,
# no divergent region) and the work-queue profile (THOR_PROF_MD_PARTS=15: all MD counters, the loop ones accumulated in registers) of the
# LDB and RA operating points.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
O=$R/gpurun_out
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s]"; }
python3 -m thor_amd.synth /tmp/w/sd.yuv 640 384 5 2
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 9 2
hang() {  # tag lib clip w h streams
  gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_$1 tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_$2.so -Wl,-rpath,$R/thor_amd
  THOR_PROF=md THOR_HIP_SPIN_TIMEOUT_S=20 timeout 50 /tmp/w/thorenc_$1 -cf $R/configs/ldb_high_efficiency.cfg -if $3 -width $4 -height $5 -qp 32 -f 30 -n 4 -streams $6 -wrap 5 > $O/r4c5_hang_$1.log 2>&1
  rc=$?; echo "$(el) hang test $1: rc=$rc $(grep -E 'thorenc_hip:|aborted|scheduler failed' $O/r4c5_hang_$1.log | cut -c1-150)"; return $rc
}
prof() {  # tag cfg n streams qp
  gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_p15 tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_profmd15.so -Wl,-rpath,$R/thor_amd
  THOR_PROF=md timeout 200 /tmp/w/thorenc_p15 -cf $R/configs/$2 -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp $5 -f 30 -n $3 -streams $4 -wrap 9 > $O/r4c5_profmd_$1.log 2>&1
  echo "$(el) profmd $1 rc=$?"; grep -v "^[WIE]2026" $O/r4c5_profmd_$1.log | grep -E "thorenc_hip:|sb_total|items|master|barrier|parked|fork|lockstep"
}
prof ldb ldb_high_efficiency.cfg 6 128 32
prof ra ra_high_efficiency.cfg 9 96 27
hang md1L md1L /tmp/w/sd.yuv 640 384 24
hang md1U md1U /tmp/w/sd.yuv 640 384 24
