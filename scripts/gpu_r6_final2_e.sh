#!/bin/bash
# round 6, second final call E: bench.main() over RCCL with ONE rank (torch.distributed.run --nproc-per-node 1; the pool hands out one GPU: no N > 1 RCCL run exists)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$R/gpurun_out
timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline > $O/r06_rccl_single_rank.json 2> $O/r6f2e.err
echo "rc=$? $(grep -o '"value": [0-9.]*' $O/r06_rccl_single_rank.json | head -1) $(grep -o '"bit_exact": [a-z]*' $O/r06_rccl_single_rank.json) $(grep -o '"n_gpus": [0-9]*' $O/r06_rccl_single_rank.json)"; tail -2 $O/r6f2e.err | cut -c1-200
