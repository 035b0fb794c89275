set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof /tmp/w
export TMPDIR=/tmp
python3 oracle/gen_clip.py /tmp/w/hd.yuv 1920 1080 3 2
ARGS="-cf configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 2 -streams 32 -wrap 3"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r01 -- tools/thorenc_hip $ARGS 2>&1 | tail -5
ls -R gpurun_out/prof | head -20
find gpurun_out/prof -name "*kernel_stats*" | head -2 | xargs -I{} sh -c 'echo {}; cat {}'
