#!/bin/bash
# round 5, final call E: bench.main() over RCCL with one rank (torch.distributed.run --nproc-per-node 1): consistency broadcast, every-rank verification
# collected with all_gather_object, ordered gather of the bitstreams, all-reduce of the totals, reference legs on rank 0 - the code path of an N-GPU run.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --width 1920 --height 1080 --streams 32 --warmup 1 --steps 4 > $O/r05_rccl_single_rank.json 2> $O/r05_rccl_single_rank.err
echo "rc=$? $(grep -o '"value": [0-9.]*' $O/r05_rccl_single_rank.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r05_rccl_single_rank.json) $(grep -o '"n_gpus": [0-9]*' $O/r05_rccl_single_rank.json) $(grep -o '"cpu_baseline": {"value": [0-9.a-z]*' $O/r05_rccl_single_rank.json)"; tail -3 $O/r05_rccl_single_rank.err
