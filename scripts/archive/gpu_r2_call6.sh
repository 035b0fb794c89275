#!/bin/bash
# round 2, call 6: LDS-resident sample blocks for small coding blocks (A/B), PMC passes on a saturated workload, profiling-build diagnostic
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
python3 -m thor_amd.synth /tmp/w/cif.yuv 416 240 4 7
( time THOR_PROF=1 THOR_HIP_SPIN_TIMEOUT_S=15 timeout 60 stdbuf -o0 -e0 tools/thorenc_hip_prof -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/cif.yuv -width 416 -height 240 -qp 32 -f 30 -n 2 -streams 1 ) > gpurun_out/r2c6_prof_small.log 2>&1
echo "prof small rc=$?"; head -4 gpurun_out/r2c6_prof_small.log
ab() {  # tag lib [env...]
  tag=$1; lib=$2; shift 2
  env "$@" THOR_HIP_LIB=$R/thor_amd/$lib timeout 300 python bench.py --width 1920 --height 1080 --streams 128 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > gpurun_out/r2c6_ab_$tag.log 2>&1
  echo "$tag: $(grep -o '"value": [0-9.]*' gpurun_out/r2c6_ab_$tag.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2c6_ab_$tag.log)"
}
ab g3 libthor_hip_g3.so X=1
ab e1 libthor_hip_e1.so X=1
ab e1b8 libthor_hip_e1b8.so X=1
ab e1_wg512 libthor_hip_e1.so THOR_HIP_WGS=512
( time THOR_HIP_LIB=$R/thor_amd/libthor_hip_e1.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x ) > gpurun_out/r2c6_tests_e1.log 2>&1
tail -3 gpurun_out/r2c6_tests_e1.log
# PMC passes on a SATURATED workload: 1080p, 128 streams, I + 2 P, library = e1
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 4 2
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_e1 tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_e1.so -Wl,-rpath,$R/thor_amd
PARGS="-cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 3 -streams 128 -wrap 4"
cd /tmp
pmc() {
  tag=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/r2c6_pmc_$tag -- /tmp/w/thorenc_e1 $PARGS > $R/gpurun_out/r2c6_pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"; grep -v "^[WIE]2026" $R/gpurun_out/r2c6_pmc_$tag.log | tail -1
}
pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
