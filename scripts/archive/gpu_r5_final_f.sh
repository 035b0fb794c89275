#!/bin/bash
# round 5, final call F: the fourth counter group of the driver's regime (vector-memory / FLAT / scalar-memory instruction counts of the 41 timed launches),
# same command as the passes of call B -> profiles/r05_pmc_bench_sq2.md
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
cd /tmp
timeout 330 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/r5pmc_sq2 -- python $R/bench.py --warmup 5 --steps 20 --verify recorded --no-cpu-baseline > $O/r5pmc_sq2.log 2>&1
echo "pmc sq2 rc=$? $(grep -o '"value": [0-9.]*' $O/r5pmc_sq2.log | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r5pmc_sq2.log)"
cd $R
python3 - <<PY
import collections, csv, glob, sys
sys.path.insert(0, '$R')
from bench import csrc_digest
fs = glob.glob('$O/r5pmc_sq2/**/*_counter_collection.csv', recursive=True)
rows = [r for r in csv.DictReader(open(fs[0])) if 'k_superblocks' in r['Kernel_Name']]
ids = sorted({int(r['Dispatch_Id']) for r in rows})
keep = set(ids[11:])
agg = collections.OrderedDict()
for r in rows:
    if int(r['Dispatch_Id']) in keep:
        agg[r['Counter_Name']] = agg.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
px = 3840.0 * 2160 * 128 * 20
L = ['# rocprofv3 PMC pass 4 (vector / scalar memory instructions) on k_superblocks, the driver\'s regime', '',
     'Command: cd /tmp && rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES --kernel-trace -- python bench.py --warmup 5 --steps 20 --verify recorded --no-cpu-baseline',
     f'{len(keep)} timed launches of k_superblocks (of {len(ids)}; the 11 launches of the warm-up frames are left out); engine sources {csrc_digest()} (= profiles/r05_pmc_bench.json).', '',
     '| counter | sum | per luma pixel of the timed frames |', '|---|---|---|']
for k, v in agg.items():
    L.append(f'| {k} | {v:.4g} | {v / px:.4g} |')
open('$O/r05_pmc_bench_sq2.md', 'w').write('\n'.join(L) + '\n')
print('\n'.join(L))
PY
find $O/r5pmc_sq2 -name "*.csv" -size +1M -delete
