#!/bin/bash
# round 3, call 1: is k_superblocks instruction-fetch bound?  PMC passes (SQC_ICACHE_*, SQ_IFETCH, TCP/TCC hit rates, memory latencies)
# on the round-2 library + A/B of code-size variants (count/emit split of the bit sink, -Os, 2 waves/SIMD)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
ab() {
  tag=$1; lib=$2; shift 2
  THOR_HIP_LIB=$R/thor_amd/$lib timeout 300 python bench.py --width 1920 --height 1080 --streams 128 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > gpurun_out/r3c1_ab_$tag.log 2>&1
  echo "$tag: $(grep -o '"value": [0-9.]*' gpurun_out/r3c1_ab_$tag.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r3c1_ab_$tag.log)"
}
ab r2 libthor_hip_r2.so
ab split libthor_hip.so
ab split_os libthor_hip_os.so
ab r2_occ2 libthor_hip_r2_occ2.so
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 4 2
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_r2 tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_r2.so -Wl,-rpath,$R/thor_amd
PARGS="-cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 3 -streams 128 -wrap 4"
cd /tmp
pmc() {
  tag=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/r3c1_pmc_$tag -- /tmp/w/thorenc_r2 $PARGS > $R/gpurun_out/r3c1_pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"; grep -v "^[WIE]2026" $R/gpurun_out/r3c1_pmc_$tag.log | tail -1
}
pmc ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_ANY
pmc lvl SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_INSTS_SMEM
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in ('ic', 'lvl', 'tcc', 'tcp'):
    fs = glob.glob(f'gpurun_out/r3c1_pmc_{tag}/*/*_counter_collection.csv')
    if not fs: print(tag, 'no csv'); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        if 'k_superblocks' in r['Kernel_Name']:
            agg[r['Counter_Name']] = agg.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    print(tag, {k: '%.4g' % v for k, v in agg.items()})
PY
