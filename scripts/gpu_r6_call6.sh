#!/bin/bash
# round 6, call 6: the 16-bit sub-pel search with eight lanes per candidate (me_cand16_subpel): parity on the 10/12-bit goldens + known answers, A/B against
# the previous commit on the 10-bit HDB16 operating point; Thorenc_hip (the reference's front end on libthor_hip.so) against Thorenc, wall clock.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py -q -m gpu > $O/r6c6_par.log 2>&1; echo "$(el) parity + kat rc=$? $(tail -1 $O/r6c6_par.log)"; grep -E "^FAILED|^ERROR" $O/r6c6_par.log | head
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "10bit or 12bit" > $O/r6c6_big.log 2>&1; echo "$(el) full-size 10/12-bit goldens rc=$? $(tail -1 $O/r6c6_big.log)"; grep -E "^FAILED|^ERROR" $O/r6c6_big.log | head
AB10="--config hdb16 --bitdepth 10 --width 1920 --height 1080 --streams 96 --warmup 1 --steps 16 --no-verify --no-cpu-baseline"
for v in pre16s new pre16s new; do
  lib=$R/thor_amd/libthor_hip_$v.so; [ $v = new ] && lib=$R/thor_amd/libthor_hip.so
  THOR_HIP_LIB=$lib timeout 400 python bench.py $AB10 > $O/r6c6_ab10_$v.log 2>$O/r6c6_ab10_$v.err
  echo "$(el) 1080p 10-bit HDB16 s96 $v: $(grep -o '"value": [0-9.]*' $O/r6c6_ab10_$v.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r6c6_ab10_$v.log)"
done
python3 -m thor_amd.synth /tmp/w/uhd.yuv 3840 2160 5 4
for b in Thorenc_hip Thorenc; do
  s0=$(date +%s.%N)
  $R/oracle/_ref/$b -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/uhd.yuv -width 3840 -height 2160 -qp 32 -f 30 -n 5 -of /tmp/w/$b.bit -rf /tmp/w/$b.yuv > $O/r6c6_$b.log 2>&1
  s1=$(date +%s.%N)
  echo "$(el) $b 3840x2160 x 5 frames (I + 4 P): $(python3 -c "print('%.1f s wall = %.3f Mpixels/s' % ($s1 - $s0, 5 * 3840 * 2160 / ($s1 - $s0) / 1e6))") rc=$?"
done
cmp /tmp/w/Thorenc_hip.bit /tmp/w/Thorenc.bit && cmp /tmp/w/Thorenc_hip.yuv /tmp/w/Thorenc.yuv && echo "Thorenc_hip == Thorenc: bitstream and reconstruction identical"
