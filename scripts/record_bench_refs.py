#!/usr/bin/env python3
"""Record what the reference encoder (oracle/_ref/Thorenc) produces for streams of bench.py's workload: bitstream md5 + size and the
md5 of every reconstructed frame (display order) -> tests/golden/bench_refs.json.  For the operating points whose reference runs are
too long to wait for on the GPU box (3840x2160 RA: ~5 min, 3840x2160 10-bit HDB16 with its 17-frame chunk: ~11 min per stream on one
core): `bench.py --verify recorded` compares the GPU's streams with these records instead of live runs.  Build container only
(`make -C oracle` first); the clips come from the seeded generator, exactly as bench.py builds them.
  python scripts/record_bench_refs.py --config ra --frames 9 --streams 96 [--width 3840 --height 2160 --bitdepth 8 --sigma 2 -j 3]"""
import argparse, hashlib, json, os, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from thor_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', choices=sorted(bench.CONFIGS), required=True)
    ap.add_argument('--frames', type=int, required=True, help='warmup + steps of the bench run')
    ap.add_argument('--streams', type=int, required=True, help='streams per GPU of the bench run (the first, middle and last one are recorded)')
    ap.add_argument('--width', type=int, default=3840)
    ap.add_argument('--height', type=int, default=2160)
    ap.add_argument('--bitdepth', type=int, default=8)
    ap.add_argument('--sigma', type=float, default=2.0)
    ap.add_argument('--qp', type=int, default=None)
    ap.add_argument('-j', type=int, default=3)
    ap.add_argument('--sids', default=None, help='comma-separated stream ids to record instead of first / middle / last')
    ap.add_argument('--clip-frames', type=int, default=None, help='generate the clip as for a run of this many frames (bench.py --clip-frames) and code the first --frames of it')
    ap.add_argument('--cpu-only', action='store_true',
                    help='record only the CPU seconds of the run (key suffix _cpu_of_n<clip frames>): the shorter leg of a recorded CPU-baseline difference')
    a = ap.parse_args()
    cfg_name, qp_default = bench.CONFIGS[a.config]
    qp = a.qp if a.qp is not None else qp_default
    w, h, n, S = a.width, a.height, a.frames, a.streams
    clipn = a.clip_frames or n
    base = synth.make_clip(w, h, clipn + bench.EXTRA_FRAMES, bench.CONTENT_SEED(a.config, w), a.sigma, a.bitdepth)
    sids = sorted({0, S // 2, S - 1}) if a.sids is None else sorted({int(x) for x in a.sids.split(',')})
    extra = ['-bitdepth', str(a.bitdepth), '-input_bitdepth', str(a.bitdepth)] if a.bitdepth > 8 else []
    fbytes = w * h * 3 // 2 * (2 if a.bitdepth > 8 else 1)
    path = os.path.join(ROOT, 'tests', 'golden', 'bench_refs.json')
    model = [l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')]
    HOST = f'{model[0] if model else "unknown CPU"}, {os.cpu_count()} logical cores (build container)'
    out = json.load(open(path)) if os.path.exists(path) else {}

    def job(sid):
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, 'in.yuv'), 'wb') as f:
                for fr in bench.host_stream_frames(base, sid, clipn, a.bitdepth)[:n]:
                    f.write(fr.tobytes())
            pr = subprocess.Popen([bench.REF_ENC, '-cf', os.path.join(ROOT, 'configs', cfg_name), '-if', os.path.join(d, 'in.yuv'), '-width', str(w), '-height', str(h),
                                   '-qp', str(qp), '-n', str(n), '-f', '30', '-of', os.path.join(d, 'o.bit'), '-rf', os.path.join(d, 'o.yuv')] + extra,
                                  stdout=subprocess.DEVNULL)
            _, status, ru = os.wait4(pr.pid, 0)   # CPU seconds of exactly this process (user + system), as bench.py's legs take them
            pr.returncode = os.waitstatus_to_exitcode(status)
            assert pr.returncode == 0
            cpu = {'cpu_s': round(ru.ru_utime + ru.ru_stime, 2), 'cpu_host': HOST, 'cpu_concurrent_jobs': min(a.j, len(sids)),
                   'cfg_md5': bench.file_md5(os.path.join(ROOT, 'configs', cfg_name)), 'ref_enc_md5': bench.file_md5(bench.REF_ENC)}
            if a.cpu_only:
                return sid, cpu
            bits = open(os.path.join(d, 'o.bit'), 'rb').read()
            rec = open(os.path.join(d, 'o.yuv'), 'rb').read()
            assert len(rec) == n * fbytes
            # md5 of the stream prefix after every coded frame (low-delay configurations: a shorter chunk of the same stream is a prefix)
            pre = [] if a.config != 'ldb' else [hashlib.md5(bench.stream_prefix(bits, k)).hexdigest() for k in range(1, n + 1)]
            return sid, dict({'bit_md5': hashlib.md5(bits).hexdigest(), 'bit_bytes': len(bits), 'prefix_md5': pre,
                              'rec_md5': [hashlib.md5(rec[i * fbytes:(i + 1) * fbytes]).hexdigest() for i in range(n)]}, **cpu)

    with ThreadPoolExecutor(a.j) as ex:
        for sid, res in ex.map(job, sids):
            import fcntl
            with open(path + '.lock', 'w') as lk:   # several recorders may run side by side: merge under a lock
                fcntl.flock(lk, fcntl.LOCK_EX)
                out = json.load(open(path)) if os.path.exists(path) else {}
                out[bench.ref_key(a.config, w, h, a.bitdepth, qp, n, a.sigma, sid) + (f'_cpu_of_n{clipn}' if a.cpu_only else '')] = res
                json.dump(out, open(path, 'w'), indent=1, sort_keys=True)
            print('recorded', bench.ref_key(a.config, w, h, a.bitdepth, qp, n, a.sigma, sid), flush=True)


if __name__ == '__main__':
    main()
