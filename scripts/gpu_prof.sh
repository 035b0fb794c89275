set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/w
python3 -m thor_amd.synth /tmp/w/hard.yuv 352 288 8 7 --sigma 6
ARGS="-cf configs/ldb_high_efficiency.cfg -if /tmp/w/hard.yuv -width 352 -height 288 -qp 32 -n 4 -f 30"
oracle/_ref/Thorenc $ARGS -of /tmp/w/ref.bit -rf /tmp/w/ref.yuv > /dev/null
THOR_PROF=1 timeout 300 tools/thorenc_hip $ARGS -of /tmp/w/my.bit -rf /tmp/w/my.yuv
cmp /tmp/w/ref.bit /tmp/w/my.bit && echo BIT_EXACT_STREAM; cmp /tmp/w/ref.yuv /tmp/w/my.yuv && echo BIT_EXACT_RECON
