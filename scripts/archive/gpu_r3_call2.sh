#!/bin/bash
# round 3, call 2: (a) instruction-cache micro-benchmark, (b) parity of the experimental variants on the small goldens,
# (c) A/B at 1080p / 128 streams in the 4-reference regime, (d) PMC passes (instruction fetch, memory levels) on the same workload
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
O=$R/gpurun_out
timeout 120 tools/ubench_icache > $O/r3c2_icache.log 2>&1; echo "icache rc=$?"; tail -50 $O/r3c2_icache.log
par() {
  tag=$1
  THOR_HIP_LIB=$R/thor_amd/libthor_hip_$tag.so timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "golden or two_streams" > $O/r3c2_par_$tag.log 2>&1
  echo "parity $tag rc=$? $(tail -1 $O/r3c2_par_$tag.log)"
}
ab() {
  tag=$1
  THOR_HIP_LIB=$R/thor_amd/libthor_hip_$tag.so timeout 300 python bench.py --width 1920 --height 1080 --streams 128 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > $O/r3c2_ab_$tag.log 2>&1
  echo "ab $tag: $(grep -o '"value": [0-9.]*' $O/r3c2_ab_$tag.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r3c2_ab_$tag.log)"
}
par base; par sync1; par sync2; par dpp; par sync1dpp
ab base; ab sync1; ab sync2; ab dpp; ab sync1dpp; ab os
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 6 2
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_base tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_base.so -Wl,-rpath,$R/thor_amd
PARGS="-cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 5 -streams 128 -wrap 6"
cd /tmp
pmc() {
  tag=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/r3c2_pmc_$tag -- /tmp/w/thorenc_base $PARGS > $O/r3c2_pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"; grep -v "^[WIE]2026" $O/r3c2_pmc_$tag.log | tail -1
}
pmc ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_ANY
pmc lvl SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in ('ic', 'lvl', 'tcc', 'tcp'):
    fs = glob.glob(f'gpurun_out/r3c2_pmc_{tag}/*/*_counter_collection.csv')
    if not fs: print(tag, 'no csv'); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        if 'k_superblocks' in r['Kernel_Name']:
            agg[r['Counter_Name']] = agg.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    print(tag, {k: '%.4g' % v for k, v in agg.items()})
PY
rm -rf $O/r3c2_pmc_*/*/*_kernel_trace.csv 2>/dev/null
du -sh $O
