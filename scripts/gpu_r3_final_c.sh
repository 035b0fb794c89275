#!/bin/bash
# round 3, final C: throughput lines of the other BASELINE configurations (parity of these operating points is in the -m gpu suite:
# 4k_ra_n9_q27, 4k_hdb16_10bit_n3_q32, hdb16_416x240_10bit_n17_q32, 1080p_ldb_n5/n14) and one run of the RCCL path on the GPU
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$R/gpurun_out
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --width 1920 --height 1080 --streams 32 --warmup 2 --steps 2 --no-cpu-baseline > $O/r3_rccl_single_rank.json 2> $O/r3_rccl_single_rank.err; echo "rccl rc=$?"; cut -c1-400 $O/r3_rccl_single_rank.json
timeout 300 python bench.py --width 1920 --height 1080 --streams 256 --warmup 5 --steps 2 --no-verify --no-cpu-baseline > $O/r3_bench_1080p_ldb.json 2> $O/r3_bench_1080p_ldb.err; echo "1080p rc=$?"; cut -c1-300 $O/r3_bench_1080p_ldb.json
timeout 500 python bench.py --config ra --streams 48 --warmup 1 --steps 8 --no-verify --no-cpu-baseline > $O/r3_bench_4k_ra.json 2> $O/r3_bench_4k_ra.err; echo "ra rc=$?"; cut -c1-300 $O/r3_bench_4k_ra.json
timeout 500 python bench.py --config hdb16 --bitdepth 10 --streams 48 --warmup 1 --steps 8 --no-verify --no-cpu-baseline > $O/r3_bench_4k_hdb16_10bit.json 2> $O/r3_bench_4k_hdb16_10bit.err; echo "hdb16 rc=$?"; cut -c1-300 $O/r3_bench_4k_hdb16_10bit.json
