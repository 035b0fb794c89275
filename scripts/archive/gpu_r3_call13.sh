#!/bin/bash
# round 3, call 13 (last GPU seconds): vector-L1 / L2 counters of the FINAL library on the 1080p / 128-stream workload of call 2 / 4
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
O=$R/gpurun_out
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 6 2
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc tools/thorenc_hip.c -Lthor_amd -lthor_hip -Wl,-rpath,$R/thor_amd
cd /tmp
timeout 100 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/r3c13_pmc -- /tmp/w/thorenc -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 5 -streams 128 -wrap 6 > $O/r3c13_pmc.log 2>&1
echo "rc=$? $(grep thorenc_hip: $O/r3c13_pmc.log | cut -c1-170)"
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.OrderedDict()
for f in glob.glob('gpurun_out/r3c13_pmc/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_superblocks' in r['Kernel_Name']:
            agg[r['Counter_Name']] = agg.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
print({k: '%.4g' % v for k, v in agg.items()})
PY
