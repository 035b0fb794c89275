set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/w
python3 oracle/gen_clip.py /tmp/w/hd.yuv 1920 1080 6 2
ARGS="-cf configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30"
# variant A: launch_bounds(64,4) (128 VGPR)
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or two_streams" 2>&1 | tail -2
timeout 600 tools/thorenc_hip $ARGS -n 3 -streams 512 -wrap 6
cp thor_amd/libthor_hip.so /tmp/w/w4.so; cp thor_amd/libthor_hip_w3.so thor_amd/libthor_hip.so
timeout 600 tools/thorenc_hip $ARGS -n 3 -streams 512 -wrap 6
