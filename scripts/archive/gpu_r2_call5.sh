#!/bin/bash
# round 2, call 5: A/B of register budgets (2/3/4 waves per SIMD) with uniform scalars + global-typed pointers at 1080p in
# the 4-reference regime; parity subset of the candidate build; PMC passes; profiling-build diagnostic
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
python3 -m thor_amd.synth /tmp/w/cif.yuv 416 240 6 7
( time THOR_PROF=1 THOR_HIP_SPIN_TIMEOUT_S=30 timeout 90 tools/thorenc_hip_prof -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/cif.yuv -width 416 -height 240 -qp 32 -f 30 -n 5 -streams 8 -wrap 6 ) > gpurun_out/r2c5_prof_small.log 2>&1
echo "prof small rc=$?"; head -2 gpurun_out/r2c5_prof_small.log
ab() {  # tag lib
  THOR_HIP_LIB=$R/thor_amd/$2 timeout 300 python bench.py --width 1920 --height 1080 --streams 128 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > gpurun_out/r2c5_ab_$1.log 2>&1
  echo "$1: $(grep -o '"value": [0-9.]*' gpurun_out/r2c5_ab_$1.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2c5_ab_$1.log)"
}
ab uni libthor_hip_uni.so
ab g3 libthor_hip_g3.so
ab g4 libthor_hip_g4.so
ab g2 libthor_hip_g2.so
( time THOR_HIP_LIB=$R/thor_amd/libthor_hip_g4.so timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py -m gpu -q -x ) > gpurun_out/r2c5_tests_g4.log 2>&1
tail -3 gpurun_out/r2c5_tests_g4.log
# PMC passes (counters only, their own runs): 1080p, 16 streams, I + 2 P, default library
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 4 2
PARGS="-cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 3 -streams 16 -wrap 4"
cd /tmp
pmc() {
  tag=$1; shift
  timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/r2c5_pmc_$tag -- $R/tools/thorenc_hip $PARGS > $R/gpurun_out/r2c5_pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"; grep -v "^[WIE]2026" $R/gpurun_out/r2c5_pmc_$tag.log | tail -2
}
pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
cd $R; find gpurun_out -path "*r2c5_pmc*" -name "*.csv" | head -20
