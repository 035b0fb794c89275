#!/bin/bash
# round 5, call 7: speculative step / round merging in the candidate-per-lane search - cross-check parity, A/B (round-4 code / first candidate-per-lane
# library / this one) on the driver-regime proxy, and the phase profile (-DTHOR_PROF) over 13 frames for the new distribution of the wave-time.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out /tmp/w; O=$R/gpurun_out
THOR_HIP_LIB=$R/thor_amd/libthor_hip_xcheck.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "gpu_matches or staggered or two_streams" > $O/r5c7_par_xcheck.log 2>&1; echo "parity xcheck rc=$? $(tail -1 $O/r5c7_par_xcheck.log)"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "six_frames_each or 1080p_ldb_n5" > $O/r5c7_par_big.log 2>&1; echo "parity big rc=$? $(tail -1 $O/r5c7_par_big.log)"
for v in base5 cand1 new; do
  lib=$R/thor_amd/libthor_hip_$v.so; [ $v = new ] && lib=$R/thor_amd/libthor_hip.so
  THOR_HIP_LIB=$lib timeout 300 python bench.py --width 1920 --height 1080 --streams 256 --warmup 5 --steps 4 --no-verify --no-cpu-baseline --lockstep > $O/r5c7_ab_$v.log 2>$O/r5c7_ab_$v.err
  echo "1080p s256 P5-P8 lockstep $v: $(grep -o '"value": [0-9.]*' $O/r5c7_ab_$v.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r5c7_ab_$v.log)"
done
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 15 2
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_prof tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_prof.so -Wl,-rpath,$R/thor_amd
THOR_PROF=1 timeout 300 /tmp/w/thorenc_prof -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 14 -streams 128 -wrap 15 > $O/r5c7_prof_ldb_n14.log 2>&1
echo "prof rc=$?"; grep -v "^[WIE]2026" $O/r5c7_prof_ldb_n14.log | tail -36
