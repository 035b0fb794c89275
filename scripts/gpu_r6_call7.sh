#!/bin/bash
# round 6, call 7: where the wave-time goes in the final sources (-DTHOR_PROF / -DTHOR_PROF_ME builds; 1920x1080 x 128 streams, I + 13 P: the workload of
# profiles/r05_prof_ldb_1080p_n14_call8.log) and on 10-bit samples (HDB16, 48 streams, 17 frames)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 15 2
python3 -m thor_amd.synth /tmp/w/hd10.yuv 1920 1080 18 5 --bits 10
for v in prof profme; do
  gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_$v tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_$v.so -Wl,-rpath,$R/thor_amd
done
THOR_HIP_KERNEL=std THOR_PROF=1 timeout 300 /tmp/w/thorenc_prof -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 14 -streams 128 -wrap 15 > $O/r06_prof_ldb_1080p_n14.log 2>&1
echo "$(el) prof ldb rc=$?"; grep -v "^[WIE]2026" $O/r06_prof_ldb_1080p_n14.log | tail -45
THOR_HIP_KERNEL=std THOR_PROF=me timeout 300 /tmp/w/thorenc_profme -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 14 -streams 128 -wrap 15 > $O/r06_prof_ldb_me_1080p_n14.log 2>&1
echo "$(el) prof me ldb rc=$?"; grep -v "^[WIE]2026" $O/r06_prof_ldb_me_1080p_n14.log | grep -E "me_cb|calls" | head -14
THOR_PROF=1 timeout 300 /tmp/w/thorenc_prof -cf $R/configs/hdb16_high_efficiency.cfg -if /tmp/w/hd10.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 17 -streams 48 -wrap 18 -bitdepth 10 -input_bitdepth 10 > $O/r06_prof_hdb16_1080p.log 2>&1
echo "$(el) prof hdb16 rc=$?"; grep -v "^[WIE]2026" $O/r06_prof_hdb16_1080p.log | tail -40
