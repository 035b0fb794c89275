#!/bin/bash
# round 6, call 9: the wavefront form of the CDEF search's distortion pass (k_cdef_mse) + the selection pass with the per-filter-block minimum hoisted:
# parity (known answers + small goldens + one 3840x2160 golden), A/B against the library built from the commit before (thor_amd/libthor_hip_pre_cdef.so) at
# 1920x1080 x 256 streams and on the default invocation, and a kernel trace of the new library (per-kernel times of the filter kernels).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
timeout 900 python -m pytest tests/test_gpu_kat.py tests/test_gpu_parity.py -q -m gpu -x > $O/r6c9_parity.log 2>&1; echo "$(el) parity rc=$? $(tail -1 $O/r6c9_parity.log)"; grep -E "^FAILED|^ERROR" $O/r6c9_parity.log | head
AB="--width 1920 --height 1080 --streams 256 --warmup 5 --steps 4 --no-cpu-baseline"
for v in pre new pre new; do
  lib=$R/thor_amd/libthor_hip_pre_cdef.so; [ $v = new ] && lib=$R/thor_amd/libthor_hip.so
  THOR_HIP_LIB=$lib timeout 400 python bench.py $AB > $O/r6c9_ab_$v.log 2>$O/r6c9_ab_$v.err
  echo "$(el) 1080p s256 P5-P8 $v: $(grep -o '"value": [0-9.]*' $O/r6c9_ab_$v.log | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r6c9_ab_$v.log) $(grep -o '"timed_region_ms_per_step": {[^}]*}' $O/r6c9_ab_$v.log)"
done
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r6c9_trace -- python $R/bench.py $AB --no-verify > $O/r6c9_trace.log 2>&1
echo "$(el) trace rc=$? $(grep -o '"value": [0-9.]*' $O/r6c9_trace.log | head -1)"
f=$(find $O/r6c9_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
# per-launch durations of the CDEF kernels by grid size (the three passes of k_cdef share a name)
t=$(find $O/r6c9_trace -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python3 - "$t" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'cdef' in n or 'deblock' in n or 'make_ref' in n or 'copy_planes' in n or 'gather' in n:
        d[(n[:40], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', ''))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, v in sorted(d.items()):
    print(k, 'n=%d avg %.3f ms max %.3f ms total %.1f ms' % (len(v), sum(v) / len(v), max(v), sum(v)))
PY
cd $R
timeout 600 python bench.py --no-cpu-baseline > $O/r6c9_default_new.json 2> $O/r6c9_default_new.err
echo "$(el) default bench new: $(grep -o '"value": [0-9.]*' $O/r6c9_default_new.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r6c9_default_new.json) $(grep -o '"timed_region_ms_per_step": {[^}]*}' $O/r6c9_default_new.json)"
find $O -name "*.csv" -size +1M -delete
