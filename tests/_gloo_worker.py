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
dist.barrier()
if rank == 0:
    print('GLOO_OK')
