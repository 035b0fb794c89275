#!/bin/bash
# round 4, call 3 (short): device parity of the generalised search window / 16-bit dot-product sub-pel / joint-search split / vector
# loops build, and phase profiles of the three operating points on it (the HDB16 one for the first time).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
O=$R/gpurun_out
L=$R/thor_amd
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s]"; }
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py -q -x -m gpu -k "gpu_matches or two_streams or kat or hierarchical" > $O/r4c3_par.log 2>&1; echo "$(el) parity rc=$? $(tail -1 $O/r4c3_par.log)"
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 9 2
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from thor_amd import synth
clip = synth.make_clip(1920, 1080, 17, 5, 2.0, 10)
open('/tmp/w/hd10.yuv', 'wb').write(b''.join(np.concatenate([p.ravel() for p in fr]).astype('<u2').tobytes() for fr in clip))
PY
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_prof tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_prof.so -Wl,-rpath,$R/thor_amd
prof() {  # tag cfg n streams qp clip [extra args]
  tag=$1; cfg=$2; n=$3; S=$4; qp=$5; clip=$6; shift 6
  THOR_PROF=1 timeout 200 /tmp/w/thorenc_prof -cf $R/configs/$cfg -if $clip -width 1920 -height 1080 -qp $qp -f 30 -n $n -streams $S -wrap $n "$@" > $O/r4c3_prof_$tag.log 2>&1
  echo "$(el) prof $tag rc=$?"; grep -v "^[WIE]2026" $O/r4c3_prof_$tag.log | grep -E "thorenc_hip:|sb_total|barrier|parked|fork|lockstep|me_fullpel|me_subpel|code_tu|pred_inter|early_skip|final|bits|cost|me_telescope|me_cands|me_hex|quant|tu_fwd|tu_inv"
}
prof ldb ldb_high_efficiency.cfg 6 128 32 /tmp/w/hd.yuv
prof ra ra_high_efficiency.cfg 9 96 27 /tmp/w/hd.yuv
prof hdb16 hdb16_high_efficiency.cfg 17 48 32 /tmp/w/hd10.yuv -bitdepth 10 -input_bitdepth 10
