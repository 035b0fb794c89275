set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc /tmp/w
export TMPDIR=/tmp
python3 oracle/gen_clip.py /tmp/w/hd.yuv 1920 1080 4 2
ARGS="-cf configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 2 -streams 256 -wrap 4"
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|TCC_[A-Za-z_0-9]+|TCP_[A-Za-z_0-9]+|GRBM_[A-Z_]+|FETCH_SIZE|WRITE_SIZE|[A-Za-z]+Busy|[A-Za-z]*Util[a-z]*|OccupancyPercent|MeanOccupancy[A-Za-z]*)\b" | sort -u | tr '\n' ' ' | head -c 6000
echo
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc -o sq1 -- tools/thorenc_hip $ARGS 2>&1 | grep thorenc_hip
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc -o sq2 -- tools/thorenc_hip $ARGS 2>&1 | grep thorenc_hip
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc -o mem1 -- tools/thorenc_hip $ARGS 2>&1 | grep thorenc_hip
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc -o mem2 -- tools/thorenc_hip $ARGS 2>&1 | grep thorenc_hip
ls -la gpurun_out/pmc | head
