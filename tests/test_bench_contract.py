"""bench.py output contract, checked on the CPU with the encoder object replaced by a stand-in (the library itself
has no CPU path): one JSON line with the driver's keys, the roofline object and - at N=1 - the cpu_baseline object."""
import io
import json
import os
import sys
import contextlib
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _FakeEncoder:
    def __init__(self, params, num_streams=1, device=0):
        self.S = num_streams
        self.calls = 0

    def stage_device(self, stream, slot, ptr):
        assert ptr != 0 and 0 <= stream < self.S

    def bitstream(self, stream):
        return b'\x00\x00\x00\x01\xff' * (stream + 1)

    def recon(self, stream):
        raise AssertionError('recon is only fetched for verification')

    def encode_staged(self, slots):
        assert len(slots) == self.S
        self.calls += 1

    def kernel_time(self):
        return 12.0, self.calls, 1.0

    def kernel_time_reset(self):
        self.calls = 0

    def close(self):
        pass


def test_bench_prints_one_contract_line(monkeypatch):
    import thor_amd
    import bench
    monkeypatch.setattr(thor_amd, 'Encoder', _FakeEncoder)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--streams', '3', '--width', '64', '--height', '64', '--steps', '2', '--warmup', '1', '--no-verify'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    lines = [l for l in buf.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'u8' and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config'] and '64x64' in d['config']['workload']
    assert d['bit_exact'] is None and d['io']['stream_bytes_total'] == 5 * (1 + 2 + 3)
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-6
    # frames 1 and 2 of a chunk reference 1 and 2 earlier frames: 1.5 * (2 + 1.5) bytes per pixel
    assert abs(r['alg_bytes_per_px'] - 5.25) < 1e-9
    if os.path.exists(os.path.join(ROOT, 'oracle', '_ref', 'Thorenc')):
        c = d['cpu_baseline']
        for k in ('value', 'unit', 'cores', 'kind', 'sample'):
            assert k in c, k
        assert c['cores'] == 1 and c['kind'] == 'reference' and c['unit'] == d['unit']
