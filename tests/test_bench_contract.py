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
        self.frame_bytes = params.width * params.height * 3 // 2 * (2 if getattr(params, 'bitdepth', 8) > 8 else 1)

    def recon_into(self, stream, ptr):   # the product binding downloads into a caller-owned (pinned) buffer; the fakes go through their recon()
        import ctypes
        b = self.recon(stream).tobytes()
        ctypes.memset(ptr, 0, self.frame_bytes)
        ctypes.memmove(ptr, b, min(len(b), self.frame_bytes))

    def stage_device(self, stream, slot, ptr):
        assert ptr != 0 and 0 <= stream < self.S

    def bitstream(self, stream):
        return b'\x00\x00\x00\x01\xff' * (stream + 1)

    def recon(self, stream):
        raise AssertionError('recon is only fetched for verification')

    def encode_staged(self, slots):
        assert len(slots) == self.S
        self.calls += 1
        self.calls_total = getattr(self, 'calls_total', 0) + 1

    def encode_run(self, nframes, on_done=None):
        for _ in range(nframes):
            self.encode_staged([self.calls] * self.S)
            if on_done is not None:
                on_done(0, self.S)

    def last_display_index(self, stream):
        return self.calls_total - 1

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


class _SlowFakeEncoder(_FakeEncoder):
    """Every lock-step frame takes a while: the reference legs of a small geometry finish long BEFORE the timed region ends
    (the situation of the driver's 25-frame run, where round 3's poll-time stamps produced 73 Mpixels/s)."""
    def encode_staged(self, slots):
        import time
        time.sleep(0.8)   # (0.4 s was not enough on a loaded host: the legs must have exited before the timed region ends)
        super().encode_staged(slots)


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, 'oracle', '_ref', 'Thorenc')), reason='oracle/_ref/Thorenc not built')
def test_cpu_baseline_is_a_single_core_rate_when_the_legs_finish_early(monkeypatch):
    import thor_amd
    import bench
    monkeypatch.setattr(thor_amd, 'Encoder', _SlowFakeEncoder)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--streams', '2', '--width', '320', '--height', '192', '--steps', '3', '--warmup', '4', '--no-verify'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    d = json.loads([l for l in buf.getvalue().splitlines() if l.strip()][0])
    c = d['cpu_baseline']
    assert c.get('error') is None, c
    assert 0.05 < c['value'] < 5.0, c            # Mpixels/s of ONE host core, not "pixels / poll jitter"
    assert 'CPU time' in c['sample'] and c['n_process']['procs'] >= 2
    assert d['io']['cpu_legs_wait_s'] < 0.3      # the legs had exited before collect() was called


def test_one_frame_baseline_rejects_broken_timing():
    import bench
    # round 3's artefact: both legs "took" the time until somebody looked
    with pytest.raises(RuntimeError):
        bench.one_frame_baseline((276.9, 150.00), (276.8, 149.99), 3840 * 2160, {'frames_lo': 6})
    v, d_cpu, _ = bench.one_frame_baseline((180.0, 175.0), (155.0, 150.0), 3840 * 2160, {'frames_lo': 6})
    assert abs(d_cpu - 25.0) < 1e-9 and 0.3 < v < 0.35


def test_stream_prefix_of_a_short_stream_is_none():
    import bench
    one = (5).to_bytes(4, 'big') + b'abcde'
    assert bench.stream_prefix(one + one, 2) == one + one and bench.stream_prefix(one, 2) is None


class _ReorderFakeEncoder(_FakeEncoder):
    def begin_sequence(self, stream, first, n, total):
        self.next = getattr(self, 'next', {})
        self.next[stream] = 0

    def next_frame(self, stream):
        self.next[stream] += 1
        return self.next[stream] - 1


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, 'oracle', '_ref', 'Thorenc')), reason='oracle/_ref/Thorenc not built')
def test_cpu_baseline_with_frame_reordering_subtracts_the_intra_frame(monkeypatch):
    import thor_amd
    import bench
    monkeypatch.setattr(thor_amd, 'Encoder', _ReorderFakeEncoder)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--config', 'ra', '--streams', '2', '--width', '192', '--height', '128', '--steps', '8', '--warmup', '1', '--no-verify'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    d = json.loads([l for l in buf.getvalue().splitlines() if l.strip()][0])
    c = d['cpu_baseline']
    assert c.get('error') is None and 0.02 < c['value'] < 5.0, c
    assert 'coded frames 1..8' in c['sample'] and '9-frame run' in c['sample'] and '1-frame run' in c['sample']


class _ReconFakeEncoder(_FakeEncoder):
    def recon(self, stream):
        import numpy as np
        return np.zeros(16, dtype=np.uint8)


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, 'oracle', '_ref', 'Thorenc')), reason='oracle/_ref/Thorenc not built')
def test_recorded_verification_rejects_a_wrong_stream(monkeypatch, tmp_path):
    """--verify recorded: three streams against reference runs recorded by scripts/record_bench_refs.py; an encoder that returns
    anything else gets bit_exact false, value 0 and exit code 1.  The CPU baseline comes from a cropped live sample."""
    import subprocess
    import thor_amd
    import bench
    refs = os.path.join(ROOT, 'tests', 'golden', 'bench_refs.json')
    keep = open(refs).read() if os.path.exists(refs) else None
    try:
        subprocess.check_call([sys.executable, os.path.join(ROOT, 'scripts', 'record_bench_refs.py'), '--config', 'ldb', '--frames', '3', '--streams', '3',
                               '--width', '192', '--height', '128'])
        rec = json.load(open(refs))
        assert all(bench.ref_key('ldb', 192, 128, 8, 32, 3, 2.0, s) in rec for s in (0, 1, 2))
        monkeypatch.setattr(thor_amd, 'Encoder', _ReconFakeEncoder)
        monkeypatch.setattr(sys, 'argv', ['bench.py', '--streams', '3', '--width', '192', '--height', '128', '--steps', '2', '--warmup', '1', '--verify', 'recorded',
                                          '--cpu-sample', '128x64'])
        monkeypatch.delenv('WORLD_SIZE', raising=False)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), pytest.raises(SystemExit) as ex:
            bench.main()
        assert ex.value.code == 1
        d = json.loads([l for l in buf.getvalue().splitlines() if l.strip()][0])
        assert d['bit_exact'] is False and d['value'] == 0.0 and 'recorded' in d['bit_exact_source']
        assert len(d['bit_exact_checked']) == 3 and d['bit_exact_checked'][0]['frames'] == 3 and d['bit_exact_checked'][0]['recorded']['ok'] is False
        assert '128x64 crop' in d['cpu_baseline']['sample'] and d['cpu_baseline'].get('error') is None
    finally:
        if keep is None:
            os.remove(refs)
        else:
            open(refs, 'w').write(keep)


def test_pmc_figures_are_attached_only_to_the_sources_they_describe(monkeypatch):
    """roofline.traffic comes from a committed PMC summary of the same engine sources (digest), geometry, operating point and content -
    a summary of other sources is named in traffic_source and NOT attached (round 3 attached counters of an older library)."""
    import thor_amd
    import bench
    name = '_test_pmc_%d.json' % os.getpid()
    path = os.path.join(ROOT, 'profiles', name)
    monkeypatch.setattr(thor_amd, 'Encoder', _FakeEncoder)
    monkeypatch.setattr(bench, 'PMC_JSON', name)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    argv = ['bench.py', '--streams', '3', '--width', '64', '--height', '64', '--steps', '2', '--warmup', '1', '--no-verify', '--no-cpu-baseline']

    def run(extra=()):
        monkeypatch.setattr(sys, 'argv', argv + list(extra))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main()
        return json.loads([l for l in buf.getvalue().splitlines() if l.strip()][0])['roofline']
    pm = {'csrc_digest': 'not-these-sources', 'width': 64, 'height': 64, 'streams': 3, 'config': 'ldb', 'workload': 'test', 'fetch_bytes_per_px': 10.0,
          'write_bytes_per_px': 5.0, 'valu_util_chip': 0.5}
    try:
        json.dump(pm, open(path, 'w'))
        r = run()
        assert r['traffic'] is None and 'other engine sources' in r['traffic_source']
        pm['same_device_code'] = [{'csrc_digest': bench.csrc_digest(), 'check': 'device assembly diffed'}]   # a later source state with identical device code
        json.dump(pm, open(path, 'w'))
        r = run()
        assert r['traffic'] == round(15.0 * 64 * 64 * 3 * 2 / 2) and 'same device code: device assembly diffed' in r['traffic_source']
        del pm['same_device_code']
        pm['csrc_digest'] = bench.csrc_digest()
        json.dump(pm, open(path, 'w'))
        r = run()
        assert r['traffic'] == round(15.0 * 64 * 64 * 3 * 2 / 2) and r['ops']['valu_util_chip'] == 0.5 and 'same device code' not in r['traffic_source']
        assert run(['--sigma', '6'])['traffic'] is None       # other content than the profiled workload
        assert run(['--streams', '4'])['traffic'] is None     # other geometry
    finally:
        os.remove(path)


class _HostsimEncoder(_FakeEncoder):
    """Stand-in that really encodes: the frames bench.py stages (host tensors on a machine without a GPU) go through the host simulation of
    the engine (tests/hostsim - test infrastructure, the same engine sources as the library), so bench.py's verification sees streams and
    reconstructions that ARE bit-exact and its positive path runs on the CPU."""
    def __init__(self, params, num_streams=1, device=0):
        super().__init__(params, num_streams, device)
        self.p = params
        self.frames = [dict() for _ in range(num_streams)]
        self.out = None
        self.coded = 0

    def stage_device(self, stream, slot, ptr):
        import ctypes
        n = self.p.width * self.p.height * 3 // 2
        self.frames[stream][slot] = ctypes.string_at(ptr, n)

    def _run(self):
        import tempfile, subprocess
        from util import build_hostsim
        sim = build_hostsim()
        self.out = []
        for s in range(self.S):
            n = len(self.frames[s])
            with tempfile.TemporaryDirectory() as d:
                open(os.path.join(d, 'in.yuv'), 'wb').write(b''.join(self.frames[s][f] for f in range(n)))
                subprocess.check_call([sim, '-cf', os.path.join(ROOT, 'configs', 'ldb_high_efficiency.cfg'), '-if', os.path.join(d, 'in.yuv'), '-width', str(self.p.width),
                                       '-height', str(self.p.height), '-qp', str(self.p.qp), '-n', str(n), '-f', '30', '-of', os.path.join(d, 'o.bit'),
                                       '-rf', os.path.join(d, 'o.yuv')], stdout=subprocess.DEVNULL)
                self.out.append((open(os.path.join(d, 'o.bit'), 'rb').read(), open(os.path.join(d, 'o.yuv'), 'rb').read()))

    def encode_staged(self, slots):
        if self.out is None:
            self._run()
        self.coded += 1
        super().encode_staged(slots)

    def recon(self, stream):
        import numpy as np
        n = self.p.width * self.p.height * 3 // 2
        return np.frombuffer(self.out[stream][1][(self.coded - 1) * n:self.coded * n], dtype=np.uint8)

    def bitstream(self, stream):
        return self.out[stream][0]


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, 'oracle', '_ref', 'Thorenc')), reason='oracle/_ref/Thorenc not built')
def test_verification_passes_on_bit_exact_streams_live_and_recorded(monkeypatch):
    """The positive path of bench.py's self-verification on the CPU: streams produced by the host simulation of the engine are compared
    with live reference runs (bitstream prefix + every frame's reconstruction, three streams) and with recorded reference runs."""
    import subprocess
    import thor_amd
    import bench
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    monkeypatch.setattr(thor_amd, 'Encoder', _HostsimEncoder)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    base = ['bench.py', '--streams', '3', '--width', '192', '--height', '128', '--steps', '2', '--warmup', '1']

    def run(extra):
        monkeypatch.setattr(sys, 'argv', base + extra)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main()
        return json.loads([l for l in buf.getvalue().splitlines() if l.strip()][0])
    d = run(['--verify', 'live'])
    assert d['bit_exact'] is True and d['value'] > 0 and 'live' in d['bit_exact_source'] and 'recorded' not in d['bit_exact_source']
    assert [c['stream'] for c in d['bit_exact_checked']] == [0, 1, 2]
    assert all(c['ok'] and c['frames'] == 3 and c['live']['recon_coded_frames_compared'] == [0, 1, 2] and c['live']['timed_coded_frames_compared'] == [1, 2]
               for c in d['bit_exact_checked'])
    c = d['cpu_baseline']
    assert c.get('error') is None and 0.02 < c['value'] < 8
    # LDB (round 6): the headline CPU figure is the TIMED frames at the BENCHED geometry on the same box (the live leg and a leg of the warm-up frames);
    # the crop sample over all timed frames is a secondary field
    assert c['geometry'] == '192x128' and c['frames'] == [1, 2]
    assert 'TIMED coded frames 1..2' in c['sample'] and 'at the benched geometry 192x128' in c['sample'] and '3-frame run' in c['sample'] and '1-frame run' in c['sample']
    assert 0.02 < c['all_timed_frames_on_a_crop']['value'] < 8 and 'ALL timed coded frames 1..2' in c['all_timed_frames_on_a_crop']['sample']
    assert 'EXCLUDED' in d['io']['inputs'] and 'MEASURED' in d['io']['inputs'] and 'h2d' in d['io'] and 'rank 0' in d['io']['rank0_host_load']
    refs = os.path.join(ROOT, 'tests', 'golden', 'bench_refs.json')
    keep = open(refs).read()
    try:
        subprocess.check_call([sys.executable, os.path.join(ROOT, 'scripts', 'record_bench_refs.py'), '--config', 'ldb', '--frames', '3', '--streams', '3',
                               '--width', '192', '--height', '128'])
        d = run(['--verify', 'recorded', '--no-cpu-baseline'])
        assert d['bit_exact'] is True and d['value'] > 0 and 'recorded' in d['bit_exact_source'] and len(d['bit_exact_checked']) == 3
        assert 'live' not in d['bit_exact_source'] and 'equal the live' not in d['bit_exact_scope']
        # the default: live runs AND the recorded runs of every stream that has one
        d = run(['--no-cpu-baseline'])
        assert d['bit_exact'] is True and 'live' in d['bit_exact_source'] and 'recorded' in d['bit_exact_source']
        assert all('live' in c and 'recorded' in c and c['recorded']['recon_coded_frames_compared'] == [0, 1, 2] for c in d['bit_exact_checked'])
        assert 'every timed frame 1..2 is covered' in d['bit_exact_scope']
        # a 5-frame record serves a 3-frame run of the same clip through its prefix hashes (--clip-frames)
        subprocess.check_call([sys.executable, os.path.join(ROOT, 'scripts', 'record_bench_refs.py'), '--config', 'ldb', '--frames', '5', '--streams', '3',
                               '--width', '192', '--height', '128', '--sids', '1'])
        d = run(['--no-cpu-baseline', '--verify', 'recorded', '--clip-frames', '5']) if False else run(['--no-cpu-baseline', '--clip-frames', '5'])
        rec = [c for c in d['bit_exact_checked'] if 'recorded' in c]
        assert d['bit_exact'] is True and [c['stream'] for c in rec] == [1] and 'prefix of the recorded 5-frame run' in d['bit_exact_scope']
        # a record made with another configuration file describes another workload: dropped
        rj = json.load(open(refs))
        for k in rj:
            if k.startswith('ldb_192x128'):
                rj[k]['cfg_md5'] = 'x'
        json.dump(rj, open(refs, 'w'))
        d = run(['--no-cpu-baseline'])
        assert d['bit_exact'] is True and 'recorded' not in d['bit_exact_source']
    finally:
        open(refs, 'w').write(keep)
