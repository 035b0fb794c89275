#!/bin/bash
# round 3, call 4: LDS search window (full-pel passes, 5-offset candidates, sub-pel) + accumulating bit writer + luma-only
# prediction in the bi-prediction search: parity, A/B against the call-3 library, phase profile, TA/TCP counters of both
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
O=$R/gpurun_out
export THOR_HIP_LIB=$R/thor_amd/libthor_hip_win.so
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "golden or two_streams" > $O/r3c4_par_small.log 2>&1; echo "parity small rc=$? $(tail -1 $O/r3c4_par_small.log)"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "1080p_ldb_n5 or six_frames" > $O/r3c4_par_big.log 2>&1; echo "parity big rc=$? $(tail -1 $O/r3c4_par_big.log)"
ab() {
  tag=$1
  THOR_HIP_LIB=$R/thor_amd/libthor_hip_$tag.so timeout 300 python bench.py --width 1920 --height 1080 --streams 128 --warmup 4 --steps 2 --no-verify --no-cpu-baseline > $O/r3c4_ab_$tag.log 2>&1
  echo "ab $tag: $(grep -o '"value": [0-9.]*' $O/r3c4_ab_$tag.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r3c4_ab_$tag.log)"
}
ab new; ab win
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 7 2
for v in winprof new win; do gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_$v tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_$v.so -Wl,-rpath,$R/thor_amd; done
THOR_PROF=1 timeout 300 /tmp/w/thorenc_winprof -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 6 -streams 128 -wrap 7 > $O/r3c4_prof.log 2>&1
echo "prof rc=$?"; cat $O/r3c4_prof.log | grep -v "^[WIE]2026" | tail -34
PARGS="-cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/hd.yuv -width 1920 -height 1080 -qp 32 -f 30 -n 5 -streams 128 -wrap 6"
cd /tmp
pmc() {
  lib=$1; tag=$2; shift 2
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/r3c4_pmc_${lib}_$tag -- /tmp/w/thorenc_$lib $PARGS > $O/r3c4_pmc_${lib}_$tag.log 2>&1
  echo "pmc $lib $tag rc=$?"; grep -v "^[WIE]2026" $O/r3c4_pmc_${lib}_$tag.log | tail -1
}
for lib in new win; do
  pmc $lib tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
  pmc $lib ta TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
  pmc $lib sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r3c4_pmc_*/')):
    fs = glob.glob(d + '*/*_counter_collection.csv')
    if not fs: print(d, 'no csv'); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        if 'k_superblocks' in r['Kernel_Name']:
            agg[r['Counter_Name']] = agg.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    print(d, {k: '%.4g' % v for k, v in agg.items()})
PY
rm -rf $O/r3c4_pmc_*/*/*_kernel_trace.csv 2>/dev/null
