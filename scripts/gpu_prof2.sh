set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/w
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 6 2
ARGS="-cf configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 3 -streams 256 -wrap 4"
THOR_PROF=1 THOR_HIP_SPIN_TIMEOUT_S=100 timeout 240 tools/thorenc_hip_prof $ARGS > gpurun_out/prof_1080p_s256.log 2>&1
tail -40 gpurun_out/prof_1080p_s256.log
