#!/bin/bash
# round 3, call 8: strip-based sub-pel search for PUs >= 512 samples + final encode from the winner snapshot (win6): parity, A/B, profile
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
O=$R/gpurun_out
export THOR_HIP_LIB=$R/thor_amd/libthor_hip_win6.so
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "golden or two_streams" > $O/r3c8_par_small.log 2>&1; echo "parity small rc=$? $(tail -1 $O/r3c8_par_small.log)"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "1080p_ldb_n5 or six_frames" > $O/r3c8_par_big.log 2>&1; echo "parity big rc=$? $(tail -1 $O/r3c8_par_big.log)"
ab() {
  tag=$1
  THOR_HIP_LIB=$R/thor_amd/libthor_hip_$tag.so timeout 300 python bench.py --width 1920 --height 1080 --streams 128 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > $O/r3c8_ab_$tag.log 2>&1
  echo "ab $tag: $(grep -o '"value": [0-9.]*' $O/r3c8_ab_$tag.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r3c8_ab_$tag.log)"
}
ab win4; ab win6
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 7 2
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_prof tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_win6prof.so -Wl,-rpath,$R/thor_amd
THOR_PROF=1 timeout 300 /tmp/w/thorenc_prof -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 6 -streams 128 -wrap 7 > $O/r3c8_profme.log 2>&1
echo "prof rc=$? (slots tu4..tu64+ = motion_estimate cycles for CB 8,16,32,64,128; tuN(n) = calls)"; cat $O/r3c8_profme.log | grep -v "^[WIE]2026" | tail -34
