#!/bin/bash
# round 5, call 1: parity of the merged per-block window + the 26/27-frame goldens, the recorded-reference verification of the headline workload
# (25-frame records, prefix mode), the A/B base line of the driver-regime proxy, and a PC-sampling attempt (rocprofv3 beta) on the search-bound frames.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out /tmp/w; O=$R/gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py -q -x -m gpu > $O/r5c1_par.log 2>&1; echo "parity rc=$? $(tail -1 $O/r5c1_par.log)"
timeout 300 python bench.py --width 1920 --height 1080 --streams 256 --warmup 5 --steps 4 --no-verify --no-cpu-baseline > $O/r5c1_ab_head.log 2>$O/r5c1_ab_head.err
echo "ab head 1080p s256 P5-P8: $(grep -o '"value": [0-9.]*' $O/r5c1_ab_head.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r5c1_ab_head.log)"
timeout 400 python bench.py --clip-frames 25 --warmup 5 --steps 2 --no-cpu-baseline > $O/r5c1_4k_w5s2.log 2>$O/r5c1_4k_w5s2.err
echo "4k w5s2 rc=$? $(grep -o '"value": [0-9.]*' $O/r5c1_4k_w5s2.log | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r5c1_4k_w5s2.log) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r5c1_4k_w5s2.log)"
# PC sampling (beta): which instructions the waves of k_superblocks sit on.  Library with line tables (-gline-tables-only, same -O3 code).
cd /tmp && export TMPDIR=/tmp
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 14 2 2>/dev/null || (cd $R && python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 14 2)
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_g $R/tools/thorenc_hip.c -L$R/thor_amd -l:libthor_hip_g.so -Wl,-rpath,$R/thor_amd
rocprofv3 -L > $O/r5c1_rocprof_L.log 2>&1; grep -i -B2 -A12 "pc.sampl" $O/r5c1_rocprof_L.log | head -60
for m in "stochastic cycles 33554432" "host_trap time 5000"; do
  set -- $m
  rm -rf /tmp/pcs_$1
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 --output-format csv -d /tmp/pcs_$1 -- \
    /tmp/w/thorenc_g -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 13 -streams 96 -wrap 14 > $O/r5c1_pcs_$1.log 2>&1
  echo "pcs $1 rc=$? $(tail -2 $O/r5c1_pcs_$1.log | cut -c1-300)"
  du -sh /tmp/pcs_$1 2>/dev/null; find /tmp/pcs_$1 -name "*.csv" | head
  python3 $R/scripts/pc_hist.py /tmp/pcs_$1 200 > $O/r5c1_pcs_$1_hist.txt 2>&1; head -30 $O/r5c1_pcs_$1_hist.txt
  if [ -s $O/r5c1_pcs_$1_hist.txt ] && grep -q "samples: [1-9]" $O/r5c1_pcs_$1_hist.txt; then break; fi
done
