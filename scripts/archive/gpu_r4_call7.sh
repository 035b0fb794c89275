#!/bin/bash
# round 4, call 7 (after the final call; profiling builds of the final sources): phase profiles of the three operating points, motion-search
# cycles by coding-block size, workgroup utilisation of the final library at 3840x2160 x 128 streams in the driver's regime.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
O=$R/gpurun_out
L=$R/thor_amd
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s]"; }
python3 -m thor_amd.synth /tmp/w/hd.yuv 1920 1080 9 2
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from thor_amd import synth
clip = synth.make_clip(1920, 1080, 17, 5, 2.0, 10)
open('/tmp/w/hd10.yuv', 'wb').write(b''.join(np.concatenate([p.ravel() for p in fr]).astype('<u2').tobytes() for fr in clip))
PY
prof() {  # tag lib mode cfg n streams qp clip [extra args]
  tag=$1; lib=$2; mode=$3; cfg=$4; n=$5; S=$6; qp=$7; clip=$8; shift 8
  gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc_$lib tools/thorenc_hip.c -Lthor_amd -l:libthor_hip_$lib.so -Wl,-rpath,$R/thor_amd
  THOR_PROF=$mode timeout 200 /tmp/w/thorenc_$lib -cf $R/configs/$cfg -if $clip -width 1920 -height 1080 -qp $qp -f 30 -n $n -streams $S -wrap $n "$@" > $O/r4c7_prof_$tag.log 2>&1
  echo "$(el) prof $tag rc=$?"; grep -v "^[WIE]2026" $O/r4c7_prof_$tag.log | grep -E "thorenc_hip:|sb_total|barrier|parked|me_fullpel|me_subpel|code_tu|me cb"
}
prof ldb prof 1 ldb_high_efficiency.cfg 6 128 32 /tmp/w/hd.yuv
prof ra prof 1 ra_high_efficiency.cfg 9 96 27 /tmp/w/hd.yuv
prof hdb16 prof 1 hdb16_high_efficiency.cfg 17 48 32 /tmp/w/hd10.yuv -bitdepth 10 -input_bitdepth 10
prof ldb_me profme me ldb_high_efficiency.cfg 6 128 32 /tmp/w/hd.yuv
THOR_SBTIMES=/tmp/w/sbt.bin timeout 400 python bench.py --warmup 5 --steps 2 --no-verify --no-cpu-baseline > $O/r4c7_sbt.log 2>&1
echo "$(el) sbtimes run: $(grep -o '"value": [0-9.]*' $O/r4c7_sbt.log | head -1)"; python scripts/sbtimes.py /tmp/w/sbt.bin 768 > $O/r4c7_sbtimes.log 2>&1; tail -4 $O/r4c7_sbtimes.log
