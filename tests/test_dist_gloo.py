"""Multi-GPU path on CPU: two processes over gloo exercise bench.py's sharding + timing reduction
(streams are partitioned across ranks, no data-path collective; barrier + max-over-ranks)."""
import os
import subprocess
import sys
import pytest
from util import ROOT


def test_bench_sharding_world_size_2():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29533', os.path.join(ROOT, 'tests', '_gloo_worker.py')],
                       capture_output=True, text=True, env=env, timeout=300)
    assert 'GLOO_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
