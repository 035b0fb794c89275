import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
parts = bench.shard_streams(10, world)
assert sum(parts) == 10 and len(parts) == world and max(parts) - min(parts) <= 1
mine = bench.stream_ids(10, world, rank)
allids = [None] * world
dist.all_gather_object(allids, mine)
assert sorted(sum(allids, [])) == list(range(10))
t = bench.reduce_max_time(1.0 + rank, dist)
assert abs(t - float(world)) < 1e-9
assert bench.reduce_sum(len(mine), dist) == 10
# input side: rank 0 owns the clip, everybody ends up with the same bytes (RCCL broadcast on the GPU box, gloo here)
import numpy as np
clip = np.arange(1000, dtype=np.uint32).view(np.uint8) if rank == 0 else None
got = bench.broadcast_clip(clip, dist)
assert got.numpy().tobytes() == np.arange(1000, dtype=np.uint32).tobytes()
# output side: ordered gather of variable-length chunk bitstreams (rank-major chunk order), incl. an empty one
mine_bits = [bytes([rank * 16 + i]) * (3 + 5 * i + rank) for i in range(2 + rank)] + [b'']
allb = bench.gather_bitstreams(mine_bits, dist)
if rank == 0:
    exp = []
    for r in range(world):
        exp += [bytes([r * 16 + i]) * (3 + 5 * i + r) for i in range(2 + r)] + [b'']
    assert allb == exp, 'chunk order / payload lost in the gather'
else:
    assert allb is None
assert bench.stream_prefix(b'\x00\x00\x00\x02ab\x00\x00\x00\x01c' + b'zz', 2) == b'\x00\x00\x00\x02ab\x00\x00\x00\x01c'
dist.barrier()
if rank == 0:
    print('GLOO_OK')
