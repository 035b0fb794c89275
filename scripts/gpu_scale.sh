set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/w
python3 oracle/gen_clip.py /tmp/w/hd.yuv 1920 1080 4 2
ARGS="-cf configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30"
( time oracle/_ref/Thorenc $ARGS -n 2 -of /tmp/w/ref.bit -rf /tmp/w/ref.yuv ) 2>&1 | tail -12
timeout 400 tools/thorenc_hip $ARGS -n 2 -streams 1 -of /tmp/w/my.bit -rf /tmp/w/my.yuv
cmp /tmp/w/ref.bit /tmp/w/my.bit && echo BIT_EXACT_STREAM; cmp /tmp/w/ref.yuv /tmp/w/my.yuv && echo BIT_EXACT_RECON
timeout 400 tools/thorenc_hip $ARGS -n 2 -streams 16 -wrap 4
timeout 400 tools/thorenc_hip $ARGS -n 2 -streams 64 -wrap 4
