set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/w
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
python3 oracle/gen_clip.py /tmp/w/hd.yuv 1920 1080 6 2
ARGS="-cf configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30"
timeout 600 tools/thorenc_hip $ARGS -n 3 -streams 128 -wrap 6
timeout 600 tools/thorenc_hip $ARGS -n 3 -streams 256 -wrap 6
timeout 600 tools/thorenc_hip $ARGS -n 3 -streams 512 -wrap 6
