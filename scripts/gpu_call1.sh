set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/w
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 6 2
ARGS="-cf configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 3 -streams 256 -wrap 4"
THOR_PROF=1 THOR_HIP_SPIN_TIMEOUT_S=100 timeout 200 tools/thorenc_hip_prof $ARGS > gpurun_out/prof3_1080p_s256.log 2>&1
head -12 gpurun_out/prof3_1080p_s256.log
( time THOR_HIP_SPIN_TIMEOUT_S=60 timeout 240 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ldb_low or ldb_medium or ldb_high or n6_q32 or ra_gop4 or hdb16 or two_streams or 1080p or dropin" ) > gpurun_out/call1b_tests.log 2>&1; tail -6 gpurun_out/call1_tests.log
