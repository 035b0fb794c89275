set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/w && rocminfo | grep -E "gfx|Compute Unit" | head -4
python3 oracle/gen_clip.py /tmp/w/hard.yuv 352 288 8 7 --sigma 6
CF=configs/ldb_high_efficiency.cfg
ARGS="-cf $CF -if /tmp/w/hard.yuv -width 352 -height 288 -qp 32 -n 6 -f 30"
( time oracle/_ref/Thorenc $ARGS -of /tmp/w/ref.bit -rf /tmp/w/ref.yuv ) 2>&1 | tail -8
( time timeout 600 tools/thorenc_hip $ARGS -of /tmp/w/my.bit -rf /tmp/w/my.yuv ) 2>&1 | tail -8
ls -la /tmp/w/*.bit; cmp /tmp/w/ref.bit /tmp/w/my.bit && echo BIT_EXACT_STREAM; cmp /tmp/w/ref.yuv /tmp/w/my.yuv && echo BIT_EXACT_RECON
lscpu | grep -E "Model name|^CPU\(s\)" 
