#!/bin/bash
# round 2, call 12: bi-prediction search of P frames spread over the waves (lock-step phase): A/B, parity, phase profile
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
ab() {
  tag=$1; lib=$2; shift 2
  THOR_HIP_LIB=$R/thor_amd/$lib timeout 300 python bench.py --width 1920 --height 1080 --streams 128 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > gpurun_out/r2c12_ab_$tag.log 2>&1
  echo "$tag: $(grep -o '"value": [0-9.]*' gpurun_out/r2c12_ab_$tag.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2c12_ab_$tag.log)"
}
ab base libthor_hip.so
ab bp libthor_hip_bp.so
( time THOR_HIP_LIB=$R/thor_amd/libthor_hip_bp.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x ) > gpurun_out/r2c12_tests_bp.log 2>&1
tail -3 gpurun_out/r2c12_tests_bp.log
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 9 2
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/t_pv6 tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_pv6.so -Wl,-rpath,$R/thor_amd
( time THOR_PROF=1 timeout 300 stdbuf -o0 -e0 /tmp/w/t_pv6 -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 6 -streams 128 -wrap 7 ) > gpurun_out/r2c12_prof_bp_1080p_s128.log 2>&1
head -8 gpurun_out/r2c12_prof_bp_1080p_s128.log; grep "md_skip_merge\|me_calls" gpurun_out/r2c12_prof_bp_1080p_s128.log
