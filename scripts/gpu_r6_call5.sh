#!/bin/bash
# round 6, call 5: the library with TWO builds of the superblock kernel (throughput: 168 VGPRs / three workgroups per CU; few streams: 256 VGPRs / two per CU,
# picked by the library when a run cannot fill more).  (a) parity suite - every 8-bit golden through both kernels - known answers, occupancy guard, the
# 64-stream and 3840x2160 LDB goldens; (b) the throughput kernel is back at three workgroups per CU (1080p x 256 streams); (c) single-stream and 8-stream
# 3840x2160 lines, verified, with a CPU baseline; (d) BASELINE config 5 line; (e) Thorenc_hip against Thorenc, wall clock.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w; O=$R/gpurun_out
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s]"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py -q -m gpu > $O/r6c5_par.log 2>&1; echo "$(el) parity (both kernels) + kat rc=$? $(tail -1 $O/r6c5_par.log)"; grep -E "^FAILED|^ERROR" $O/r6c5_par.log | head
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "six_frames_each or staggered or 4k_ldb_n6 or 1080p_ldb_n5" > $O/r6c5_big.log 2>&1; echo "$(el) 64-stream + 4K LDB goldens rc=$? $(tail -1 $O/r6c5_big.log)"; grep -E "^FAILED|^ERROR" $O/r6c5_big.log | head
AB8="--width 1920 --height 1080 --streams 256 --warmup 5 --steps 4 --no-verify --no-cpu-baseline --lockstep"
timeout 400 python bench.py $AB8 > $O/r6c5_ab8_new.log 2>$O/r6c5_ab8_new.err
echo "$(el) 1080p 8-bit LDB s256 lockstep (throughput kernel): $(grep -o '"value": [0-9.]*' $O/r6c5_ab8_new.log | head -1) $(grep -o '"avg_launch_ms": [0-9.]*' $O/r6c5_ab8_new.log) $(grep -o '"superblock_kernel": {[^}]*}' $O/r6c5_ab8_new.log)"
for k in std lat; do
  THOR_HIP_KERNEL=$k timeout 400 python bench.py --streams 1 --warmup 5 --steps 20 --verify recorded --no-cpu-baseline > $O/r6c5_s1_$k.json 2> $O/r6c5_s1_$k.err
  echo "$(el) 4K LDB 1 stream, kernel $k: $(grep -o '"value": [0-9.]*' $O/r6c5_s1_$k.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r6c5_s1_$k.json)"
  THOR_HIP_KERNEL=$k timeout 400 python bench.py --streams 8 --warmup 5 --steps 20 --no-verify --no-cpu-baseline > $O/r6c5_s8_$k.json 2> $O/r6c5_s8_$k.err
  echo "$(el) 4K LDB 8 streams, kernel $k: $(grep -o '"value": [0-9.]*' $O/r6c5_s8_$k.json | head -1)"
done
timeout 900 python bench.py --streams 1 --warmup 5 --steps 20 > $O/r06_bench_4k_ldb_s1.json 2> $O/r6c5_s1.err
echo "$(el) 4K LDB 1 stream (verified line): $(grep -o '"value": [0-9.]*' $O/r06_bench_4k_ldb_s1.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r06_bench_4k_ldb_s1.json) cpu: $(python3 -c "import json;d=json.loads(open('$O/r06_bench_4k_ldb_s1.json').read().strip().splitlines()[-1]);c=d['cpu_baseline'];print(c['value'], c.get('geometry'), c.get('frames'), d['io']['h2d'])")"
timeout 900 python bench.py --streams 8 --warmup 5 --steps 20 > $O/r06_bench_4k_ldb_s8.json 2> $O/r6c5_s8.err
echo "$(el) 4K LDB 8 streams (verified line): $(grep -o '"value": [0-9.]*' $O/r06_bench_4k_ldb_s8.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r06_bench_4k_ldb_s8.json)"
timeout 900 python bench.py --config hdb16 --bitdepth 10 --streams 96 --warmup 1 --steps 16 --verify recorded --cpu-sample 1920x1080 > $O/r06_bench_4k_hdb16_10bit.json 2> $O/r6c5_cfg5.err
echo "$(el) config 5 (4K 10-bit HDB16, 96 streams): $(grep -o '"value": [0-9.]*' $O/r06_bench_4k_hdb16_10bit.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r06_bench_4k_hdb16_10bit.json)"
python3 -m thor_amd.synth /tmp/w/uhd.yuv 3840 2160 5 4
for b in Thorenc_hip Thorenc; do
  /usr/bin/time -f "%e s wall, %U s user" -o $O/r6c5_time_$b.txt $R/oracle/_ref/$b -cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/uhd.yuv -width 3840 -height 2160 -qp 32 -f 30 -n 5 -of /tmp/w/$b.bit -rf /tmp/w/$b.yuv > $O/r6c5_$b.log 2>&1
  echo "$(el) $b 3840x2160 x 5 frames: $(cat $O/r6c5_time_$b.txt)"
done
cmp /tmp/w/Thorenc_hip.bit /tmp/w/Thorenc.bit && cmp /tmp/w/Thorenc_hip.yuv /tmp/w/Thorenc.yuv && echo "Thorenc_hip == Thorenc: bitstream and reconstruction identical"
du -sh $O | tail -1
