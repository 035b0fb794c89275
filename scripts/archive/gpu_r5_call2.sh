#!/bin/bash
# round 5, call 2: the two-group schedule (thor_hip_encode_staged_run) on the MI355X - parity (small + 64-stream 1080p set), then A/B against the
# lock-step path on the driver-regime proxy (1080p x 256, frames 5..8) and at 3840x2160 x 128 (frames 5..8, every frame of 6 streams verified
# against the recorded reference runs), and the per-launch workgroup utilisation (THOR_SBTIMES).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; O=$R/gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "staggered or two_streams" > $O/r5c2_par.log 2>&1; echo "parity small rc=$? $(tail -1 $O/r5c2_par.log)"
timeout 400 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "staggered" > $O/r5c2_par64.log 2>&1; echo "parity 64 streams staggered rc=$? $(tail -1 $O/r5c2_par64.log)"
for v in lockstep stagger; do
  f=""; [ $v = lockstep ] && f="--lockstep"
  timeout 300 python bench.py --width 1920 --height 1080 --streams 256 --warmup 5 --steps 4 --no-verify --no-cpu-baseline $f > $O/r5c2_1080_$v.log 2>$O/r5c2_1080_$v.err
  echo "1080p s256 P5-P8 $v: $(grep -o '"value": [0-9.]*' $O/r5c2_1080_$v.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r5c2_1080_$v.log) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r5c2_1080_$v.log) $(grep -o '"launches": [0-9]*' $O/r5c2_1080_$v.log)"
done
for v in stagger lockstep; do
  f=""; [ $v = lockstep ] && f="--lockstep"
  timeout 500 python bench.py --clip-frames 25 --warmup 5 --steps 4 --verify recorded --no-cpu-baseline $f > $O/r5c2_4k_$v.log 2>$O/r5c2_4k_$v.err
  echo "4k s128 P5-P8 $v rc=$?: $(grep -o '"value": [0-9.]*' $O/r5c2_4k_$v.log | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r5c2_4k_$v.log) $(grep -o '"ms_per_step": [0-9.]*' $O/r5c2_4k_$v.log) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r5c2_4k_$v.log)"
done
THOR_SBTIMES=/tmp/sbt.bin timeout 500 python bench.py --clip-frames 25 --warmup 5 --steps 2 --no-verify --no-cpu-baseline > $O/r5c2_4k_sbt.log 2>&1
python scripts/sbtimes.py /tmp/sbt.bin 768 > $O/r5c2_sbtimes_4k_stagger.log 2>&1; tail -8 $O/r5c2_sbtimes_4k_stagger.log
