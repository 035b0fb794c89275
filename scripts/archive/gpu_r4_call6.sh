#!/bin/bash
# round 4, call 6 (after the final call; same library): the RCCL path of bench.py on one GPU (torch.distributed.run with one rank: consistency
# broadcast, ordered gather, all-reduce; rank 0 runs the reference legs) and the default `python bench.py` line.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$R/gpurun_out
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --width 1920 --height 1080 --streams 32 --warmup 1 --steps 4 > $O/r04_rccl_single_rank.json 2> $O/r04_rccl_single_rank.err
echo "rccl rc=$? $(grep -o '"value": [0-9.]*' $O/r04_rccl_single_rank.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r04_rccl_single_rank.json) $(grep -o '"cpu_baseline": {"value": [0-9.a-z]*' $O/r04_rccl_single_rank.json) $(grep -o '"parallelism": "[a-z0-9 -]*"' $O/r04_rccl_single_rank.json)"; tail -2 $O/r04_rccl_single_rank.err
timeout 500 python bench.py > $O/r04_bench_default.json 2> $O/r04_bench_default.err
echo "default rc=$? $(grep -o '"value": [0-9.]*' $O/r04_bench_default.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/r04_bench_default.json) $(grep -o '"bit_exact": [a-z]*' $O/r04_bench_default.json) $(grep -o '"cpu_baseline": {"value": [0-9.a-z]*' $O/r04_bench_default.json)"
