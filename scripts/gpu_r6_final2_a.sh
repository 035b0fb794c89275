#!/bin/bash
# round 6, second final call A (after the CDEF wavefront form and the third build of the kernel): the whole GPU suite on the final library, the default bench
# invocation, the other BASELINE configurations as verified lines (same commands as rounds 4 / 5 and as scripts/gpu_r6_final_a.sh), the few-stream lines with a
# CPU baseline, and the drop-in (the reference's front end linked against libthor_hip.so) against the reference encoder, wall clock.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
line() { echo "$(grep -o '"value": [0-9.]*' $1 | head -1) $(grep -o '"bit_exact": [a-z]*' $1)"; }
timeout 1800 python -m pytest tests -q -m gpu > $O/r06_gpu_suite_final.log 2>&1; echo "$(el) pytest -m gpu rc=$? $(tail -1 $O/r06_gpu_suite_final.log)"; grep -E "^FAILED|^ERROR" $O/r06_gpu_suite_final.log | head
timeout 900 python bench.py > $O/r06_bench_default.json 2> $O/r6f2a_default.err
echo "$(el) default bench: $(line $O/r06_bench_default.json)"
timeout 900 python bench.py --streams 1 --warmup 5 --steps 20 > $O/r06_bench_4k_ldb_s1.json 2> $O/r6f2a_s1.err
echo "$(el) 4K LDB 1 stream: $(line $O/r06_bench_4k_ldb_s1.json) $(grep -o '"superblock_kernel": {[^}]*}' $O/r06_bench_4k_ldb_s1.json | cut -c1-150)"
python3 -m thor_amd.synth /tmp/w/uhd.yuv 3840 2160 5 4
for b in Thorenc_hip Thorenc; do
  s0=$(date +%s.%N)
  $R/oracle/_ref/$b -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/uhd.yuv -width 3840 -height 2160 -qp 32 -f 30 -n 5 -of /tmp/w/$b.bit -rf /tmp/w/$b.yuv > $O/r6f2a_$b.log 2>&1
  s1=$(date +%s.%N)
  echo "$(el) $b 3840x2160 x 5 frames (I + 4 P): $(python3 -c "print('%.1f s wall = %.3f Mpixels/s' % ($s1 - $s0, 5 * 3840 * 2160 / ($s1 - $s0) / 1e6))") rc=$?"
done
cmp /tmp/w/Thorenc_hip.bit /tmp/w/Thorenc.bit && cmp /tmp/w/Thorenc_hip.yuv /tmp/w/Thorenc.yuv && echo "Thorenc_hip == Thorenc: bitstream and reconstruction identical"
timeout 900 python bench.py --width 1920 --height 1080 --streams 256 --warmup 5 --steps 8 > $O/r06_bench_1080p_ldb.json 2> $O/r6f2a_1080p.err
echo "$(el) config 2 (1080p LDB, 256 streams): $(line $O/r06_bench_1080p_ldb.json)"
timeout 900 python bench.py --config ra --streams 96 --warmup 1 --steps 8 --verify recorded --cpu-sample 1920x1080 > $O/r06_bench_4k_ra.json 2> $O/r6f2a_ra.err
echo "$(el) config 3 (4K RA qp 27, 96 streams): $(line $O/r06_bench_4k_ra.json)"
timeout 900 python bench.py --config hdb16 --bitdepth 10 --streams 96 --warmup 1 --steps 16 --verify recorded --cpu-sample 1920x1080 > $O/r06_bench_4k_hdb16_10bit.json 2> $O/r6f2a_cfg5.err
echo "$(el) config 5 (4K 10-bit HDB16, 96 streams): $(line $O/r06_bench_4k_hdb16_10bit.json)"
timeout 900 python bench.py --sigma 6 --streams 128 --warmup 5 --steps 2 --verify recorded --cpu-sample 1920x1080 > $O/r06_bench_sigma6.json 2> $O/r6f2a_sigma6.err
echo "$(el) hard content (sigma 6, 4K LDB): $(line $O/r06_bench_sigma6.json)"
timeout 900 python bench.py --streams 8 --warmup 5 --steps 20 --no-cpu-baseline > $O/r06_bench_4k_ldb_s8.json 2> $O/r6f2a_s8.err
echo "$(el) 4K LDB 8 streams: $(line $O/r06_bench_4k_ldb_s8.json)"
du -sh $O | tail -1
