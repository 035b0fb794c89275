#!/bin/bash
# round 2, call 4: parity of the build with device-side frame interpolation + deblock KAT, parity of the THOR_EXP_UNIFORM build,
# phase profiles (THOR_PROF builds) of the 4-reference regime at 1080p, first PMC passes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py tests/test_gpu_fullsize.py -m gpu -q --durations=8 ) > gpurun_out/r2c4_tests.log 2>&1
tail -14 gpurun_out/r2c4_tests.log
( time THOR_HIP_LIB=$PWD/thor_amd/libthor_hip_uni.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "golden or 64_streams or two_streams or live" ) > gpurun_out/r2c4_tests_uni.log 2>&1
tail -4 gpurun_out/r2c4_tests_uni.log
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 9 2
ARGS="-cf configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 6 -streams 128 -wrap 7"
THOR_PROF=1 timeout 300 tools/thorenc_hip_prof $ARGS > gpurun_out/r2c4_prof_1080p_s128.log 2>&1
head -3 gpurun_out/r2c4_prof_1080p_s128.log
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_prof_uni tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_prof_uni.so -Wl,-rpath,$PWD/thor_amd
THOR_PROF=1 timeout 300 /tmp/w/thorenc_prof_uni $ARGS > gpurun_out/r2c4_prof_uni_1080p_s128.log 2>&1
head -3 gpurun_out/r2c4_prof_uni_1080p_s128.log
# PMC passes (counters only, their own runs): small workload, 1080p, 16 streams, I + 2 P
PARGS="-cf configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 3 -streams 16 -wrap 7"
cd /tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r2c4_counters.txt 2>&1
pmc() {  # tag counters...
  tag=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2c4_pmc_$tag -- $GRAFT_REPO_ROOT/tools/thorenc_hip $PARGS > $GRAFT_REPO_ROOT/gpurun_out/r2c4_pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"; tail -2 $GRAFT_REPO_ROOT/gpurun_out/r2c4_pmc_$tag.log
}
pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
cd $GRAFT_REPO_ROOT; find gpurun_out -path "*r2c4_pmc*" -name "*.csv" | head -20; du -sh gpurun_out
