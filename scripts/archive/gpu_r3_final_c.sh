#!/bin/bash
# round 3, final C: (1) PMC passes (SQ / FETCH_SIZE / WRITE_SIZE) of the benched geometry through tools/thorenc_hip (the first attempt in
# final B generated its input from the wrong directory), (2) throughput lines of the other BASELINE configurations (parity of these
# operating points is in the -m gpu suite: 4k_ra_n9_q27, 4k_hdb16_10bit_n3_q32, hdb16_416x240_10bit_n17_q32, 1080p_ldb_n5/n14),
# (3) one run of the RCCL path on the GPU
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/w
O=$R/gpurun_out
python3 -m thor_amd.synth /tmp/w/uhd.yuv 3840 2160 7 4; ls -la /tmp/w/uhd.yuv
gcc -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -o /tmp/w/thorenc $R/tools/thorenc_hip.c -L$R/thor_amd -lthor_hip -Wl,-rpath,$R/thor_amd
PARGS="-cf $R/configs/ldb_high_efficiency.cfg -if /tmp/w/uhd.yuv -width 3840 -height 2160 -qp 32 -f 30 -n 6 -streams 128 -wrap 7"
cd /tmp
pmc() {
  tag=$1; shift
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/r3pmc_$tag -- /tmp/w/thorenc $PARGS > $O/r3pmc_$tag.log 2>&1
  echo "pmc $tag rc=$? $(grep thorenc_hip: $O/r3pmc_$tag.log | cut -c1-160)"
}
pmc sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
cd $R
python3 scripts/pmc_summary.py gpurun_out/r3pmc 3840 2160 128 6 gpurun_out/r03_pmc_bench "3840x2160 LDB_high_efficiency qp 32, 128 closed streams x (I + 5 P; P4 and P5 search 4 references) through tools/thorenc_hip, final round-3 library" | tail -9
find $O -name "*_kernel_trace.csv" -path "*r3pmc*" -size +2M -delete
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --width 1920 --height 1080 --streams 32 --warmup 2 --steps 2 --no-cpu-baseline > $O/r3_rccl_single_rank.json 2> $O/r3_rccl_single_rank.err; echo "rccl rc=$?"; cut -c1-400 $O/r3_rccl_single_rank.json
timeout 300 python bench.py --width 1920 --height 1080 --streams 256 --warmup 5 --steps 2 --no-verify --no-cpu-baseline > $O/r3_bench_1080p_ldb.json 2> $O/r3_bench_1080p_ldb.err; echo "1080p rc=$?"; cut -c1-300 $O/r3_bench_1080p_ldb.json
timeout 400 python bench.py --config ra --streams 48 --warmup 1 --steps 8 --no-verify --no-cpu-baseline > $O/r3_bench_4k_ra.json 2> $O/r3_bench_4k_ra.err; echo "ra rc=$?"; cut -c1-300 $O/r3_bench_4k_ra.json
timeout 400 python bench.py --config hdb16 --bitdepth 10 --streams 48 --warmup 1 --steps 8 --no-verify --no-cpu-baseline > $O/r3_bench_4k_hdb16_10bit.json 2> $O/r3_bench_4k_hdb16_10bit.err; echo "hdb16 rc=$?"; cut -c1-300 $O/r3_bench_4k_hdb16_10bit.json
