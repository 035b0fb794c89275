#!/bin/bash
# round 3, call 14: vector-L1 micro-benchmark of the motion search's gather patterns, plain and under rocprofv3 --pmc
# (TCP accesses and vector-memory instructions per launch -> accesses per wave-instruction of every pattern)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$R/gpurun_out
timeout 20 tools/ubench_l1gather > $O/r3_ubench_l1gather.log 2>&1
cd /tmp
timeout 45 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/r3c14_pmc -- $R/tools/ubench_l1gather > $O/r3c14_pmc.log 2>&1
echo "rc=$?"
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.OrderedDict()
for f in glob.glob('gpurun_out/r3c14_pmc/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:40], r['Grid_Size'] if 'Grid_Size' in r else '')
        agg.setdefault(k, collections.OrderedDict())
        agg[k][r['Counter_Name']] = agg[k].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
for k, v in agg.items():
    a, i = v.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0), v.get('SQ_INSTS_VMEM_RD', 0)
    print(k, {n: '%.4g' % x for n, x in v.items()}, 'accesses per instruction %.2f' % (a / i if i else 0))
PY
cat $O/r3_ubench_l1gather.log
