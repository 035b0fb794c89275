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


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, 'oracle', '_ref', 'Thorenc')), reason='oracle/_ref/Thorenc not built')
def test_bench_main_world_size_2():
    """bench.main() end to end with two ranks over gloo (the N > 1 path the driver launches on the 8-GPU node; no N > 1 RCCL run exists): the line of
    rank 0 carries roofline (cpu_baseline is an N = 1 figure), the chunk bitstreams arrive in global chunk order (asserted inside bench.py against rank 0's own), the
    totals are sums over ranks, and EVERY rank verifies the streams of its own that have a recorded reference run - rank 1's first stream included."""
    import json
    refs = os.path.join(ROOT, 'tests', 'golden', 'bench_refs.json')
    keep = open(refs).read()
    try:
        subprocess.check_call([sys.executable, os.path.join(ROOT, 'scripts', 'record_bench_refs.py'), '--config', 'ldb', '--frames', '3', '--streams', '4',
                               '--width', '192', '--height', '128', '--sids', '0,2'])
        def run(extra, port):
            env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), THOR_TEST_BENCH_ARGS=extra)
            r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                                '--master-port', str(port), os.path.join(ROOT, 'tests', '_gloo_bench_worker.py')], capture_output=True, text=True, env=env, timeout=600)
            lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
            assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-3000:]
            return json.loads(lines[0])
        # default at N > 1 (round 6): recorded reference runs verify every rank; NO reference process runs beside the timed region and no cpu_baseline
        # is reported (it is an N = 1 figure)
        d = run('', 29534)
        assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['config']['parallelism'] == 'stream-sharded x2' and d['value'] > 0
        assert d['roofline']['launches'] == 2 and 'cpu_baseline' not in d and 'no reference process' in d['io']['rank0_host_load']
        assert d['io']['frames_total'] == 4 * 3 and d['bit_exact'] is True
        by = {(c['rank'], c['stream']): c for c in d['bit_exact_checked']}
        assert (1, 2) in by and by[(1, 2)]['recorded']['ok'] and by[(1, 2)]['recorded']['recon_coded_frames_compared'] == [0, 1, 2]
        assert 'recorded' in by[(0, 0)] and 'live' not in by[(0, 0)] and 'ranks [0, 1]' in d['bit_exact_scope']
        # asked for explicitly, the live legs still run on rank 0 (the fall-back for workloads nobody recorded)
        d = run('--verify live', 29536)
        by = {(c['rank'], c['stream']): c for c in d['bit_exact_checked']}
        assert d['bit_exact'] is True and 'live' in by[(0, 0)] and 'live' in by[(0, 1)] and 'cpu_baseline' not in d
    finally:
        open(refs, 'w').write(keep)
