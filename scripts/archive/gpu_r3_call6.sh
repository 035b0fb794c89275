#!/bin/bash
# round 3, call 6: dot2 transform stages (win4): parity + KAT + A/B; workgroup utilisation (THOR_SBTIMES) at 3840x2160 with 96 / 144 streams
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
O=$R/gpurun_out
export THOR_HIP_LIB=$R/thor_amd/libthor_hip_win4.so
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "golden or two_streams" > $O/r3c6_par_small.log 2>&1; echo "parity small rc=$? $(tail -1 $O/r3c6_par_small.log)"
timeout 300 python -m pytest tests/test_gpu_kat.py -q -x -m gpu > $O/r3c6_kat.log 2>&1; echo "kat rc=$? $(tail -1 $O/r3c6_kat.log)"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "1080p_ldb_n5 or six_frames" > $O/r3c6_par_big.log 2>&1; echo "parity big rc=$? $(tail -1 $O/r3c6_par_big.log)"
ab() {
  tag=$1
  THOR_HIP_LIB=$R/thor_amd/libthor_hip_$tag.so timeout 300 python bench.py --width 1920 --height 1080 --streams 128 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > $O/r3c6_ab_$tag.log 2>&1
  echo "ab $tag: $(grep -o '"value": [0-9.]*' $O/r3c6_ab_$tag.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r3c6_ab_$tag.log)"
}
ab win3; ab win4
python3 -m thor_amd.synth /tmp/w/uhd.yuv 3840 2160 7 4
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_win4.so -Wl,-rpath,$R/thor_amd
for S in 96 144; do
  THOR_SBTIMES=/tmp/w/sbt_$S.bin timeout 400 /tmp/w/thorenc -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/uhd.yuv -width 3840 -height 2160 -qp 32 -f 30 -n 6 -streams $S -wrap 7 > $O/r3c6_sbt_$S.log 2>&1
  echo "sbtimes S=$S rc=$? $(grep thorenc_hip $O/r3c6_sbt_$S.log)"
  python3 scripts/sbtimes.py /tmp/w/sbt_$S.bin 768 >> $O/r3c6_sbt_$S.log 2>&1
  grep "^frame" $O/r3c6_sbt_$S.log | cut -c1-230
done
