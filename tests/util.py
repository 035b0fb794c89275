"""Shared helpers for the test-suite (test infrastructure; may use oracle/)."""
import ctypes as C
import gzip
import hashlib
import json
import os
import sys
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
CFG = os.path.join(ROOT, 'configs', 'ldb_high_efficiency.cfg')
REF_ENC = os.path.join(ROOT, 'oracle', '_ref', 'Thorenc')
REF_DEC = os.path.join(ROOT, 'oracle', '_ref', 'Thordec')
REF_HIPENC = os.path.join(ROOT, 'oracle', '_ref', 'Thorenc_hip')
HAVE_REFERENCE_TREE = os.path.exists('/root/reference/enc/mainenc.c')


def md5(b):
    return hashlib.md5(b).hexdigest()


def golden_streams():
    return json.load(open(os.path.join(GOLD, 'streams.json')))


def golden_clip(name):
    """Committed clip, 'gen:w,h,frames,seed,sigma[,bits]' = the seeded synthetic generator (thor_amd/synth.py), or
    'stream:w,h,frames,seed,sigma,sid,extra' = stream `sid` of the multi-stream workload of bench.py
    (synth.make_stream_frames over a base clip of frames+extra frames)."""
    if name.startswith('gen:'):
        from thor_amd import synth as gen_clip
        a = name[4:].split(',')
        w, h, n, seed, sigma = a[:5]
        bits = int(a[5]) if len(a) > 5 else 8
        return b''.join(p.tobytes() for fr in gen_clip.make_clip(int(w), int(h), int(n), int(seed), float(sigma), bits) for p in fr)
    if name.startswith('stream:'):
        from thor_amd import synth
        w, h, n, seed, sigma, sid, extra = name[7:].split(',')
        base = _base_clip(int(w), int(h), int(n) + int(extra), int(seed), float(sigma))
        return b''.join(f.tobytes() for f in synth.make_stream_frames(base, int(sid), int(n)))
    return gzip.open(os.path.join(GOLD, name)).read()


_BASE = {}


def _base_clip(w, h, n, seed, sigma):
    from thor_amd import synth
    k = (w, h, n, seed, sigma)
    if k not in _BASE:
        _BASE.clear()
        _BASE[k] = synth.make_clip(w, h, n, seed, sigma)
    return _BASE[k]


def build_oracle_c():
    out = os.path.join(ROOT, 'oracle', 'libthor_oracle.so')
    src = os.path.join(ROOT, 'oracle', 'thor_oracle.c')
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O2', '-std=c99', '-fPIC', '-shared', '-ffp-contract=off', '-o', out, src])
    return C.CDLL(out)


def build_hostsim(lanes=1, waves=1):
    """Host simulation of the engine sources (tests only): 1-lane teams, `lanes` OS threads per team, or `waves`
    wavefronts (OS threads, 1-lane teams) per workgroup."""
    assert lanes == 1 or waves == 1
    out = os.path.join(ROOT, 'tests', 'hostsim', f'hostsim_w{waves}' if waves > 1 else ('hostsim' if lanes == 1 else f'hostsim_l{lanes}'))
    src = os.path.join(ROOT, 'tests', 'hostsim', 'hostsim.cpp')
    csrc = os.path.join(ROOT, 'thor_amd', 'csrc')
    newest = max([os.path.getmtime(src)] + [os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc)])
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        extra = ['-pthread'] + ([] if lanes == 1 else [f'-DTHOR_HOSTSIM_LANES={lanes}']) + ([] if waves == 1 else [f'-DTHOR_HOSTSIM_WAVES={waves}'])
        # -fno-strict-aliasing: the host branches of the engine headers read row segments (uint32_t[4]) through sample-typed pointers (tk_me.h:seg_sad);
        # with 16-bit samples that is undefined under strict aliasing, and g++ -O2 miscompiled one instantiation in builds with 32+ lanes (round 5:
        # the 10-bit HDB16 golden differed with 32 / 64-lane teams, bit-exact with this flag; the device code works on the dwords - v_sad_u16)
        subprocess.check_call(['g++', '-std=c++17', '-O2', '-fno-strict-aliasing', '-DTHOR_HOSTSIM', '-ffp-contract=off'] + extra + ['-o', out, src])
    return out


def run_encoder(binary, clip_bytes, w, h, n, qp, extra=(), env=None, cfg=None):
    """Run a Thorenc-compatible CLI (reference, hostsim, thorenc_hip, Thorenc_hip); returns (bits, recon)."""
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 'in.yuv'), 'wb').write(clip_bytes)
        cmd = [binary, '-cf', os.path.join(ROOT, 'configs', cfg) if cfg else CFG, '-if', os.path.join(d, 'in.yuv'), '-width', str(w), '-height', str(h), '-qp', str(qp),
               '-n', str(n), '-f', '30', '-of', os.path.join(d, 'o.bit'), '-rf', os.path.join(d, 'o.yuv')] + list(extra)
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, env=env)
        return open(os.path.join(d, 'o.bit'), 'rb').read(), open(os.path.join(d, 'o.yuv'), 'rb').read()


def decode(bits):
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 's.bit'), 'wb').write(bits)
        subprocess.run([REF_DEC, os.path.join(d, 's.bit'), os.path.join(d, 'o.yuv')], check=True, stdout=subprocess.DEVNULL)
        return open(os.path.join(d, 'o.yuv'), 'rb').read()


def vp(a):
    return a.ctypes.data_as(C.c_void_p)
