#!/usr/bin/env python3
"""Dev tool: run the reference encoder and the 1-lane host simulation of the engine on a matrix of
synthetic clips / QPs / geometries and report any stream or recon mismatch.
Usage: parity_matrix.py [--jobs 8] [--out /tmp/pm] [--quick]"""
import argparse, itertools, os, subprocess, sys, hashlib
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from thor_amd import synth as gen_clip

def run_case(case, out, sim):
    w, h, frames, seed, sigma, qp, extra = case[:7]
    cfgname = case[7] if len(case) > 7 else 'ldb_high_efficiency.cfg'
    tag = f"{w}x{h}_f{frames}_s{seed}_n{sigma}_q{qp}_{os.path.basename(cfgname).split('.')[0]}_{hashlib.md5(' '.join(extra).encode()).hexdigest()[:6]}"
    clip = os.path.join(out, f"clip_{w}x{h}_{frames}_{seed}_{sigma}.yuv")
    if not os.path.exists(clip):
        gen_clip.write_clip(clip + '.tmp', gen_clip.make_clip(w, h, frames, seed, sigma))
        os.replace(clip + '.tmp', clip)
    cfg = cfgname if os.path.isabs(cfgname) else os.path.join(ROOT, 'configs', cfgname)
    args = ['-cf', cfg, '-if', clip, '-width', str(w), '-height', str(h), '-qp', str(qp), '-n', str(frames), '-f', '30'] + list(extra)
    r = subprocess.run([os.path.join(ROOT, 'oracle/_ref/Thorenc')] + args + ['-of', f'{out}/{tag}.ref.bit', '-rf', f'{out}/{tag}.ref.yuv'],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    s = subprocess.run([sim] + args + ['-of', f'{out}/{tag}.my.bit', '-rf', f'{out}/{tag}.my.yuv'],
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    if r.returncode or s.returncode:
        return tag, f'RC ref={r.returncode} sim={s.returncode} {s.stderr[-200:]}'
    same_b = open(f'{out}/{tag}.ref.bit', 'rb').read() == open(f'{out}/{tag}.my.bit', 'rb').read()
    same_y = open(f'{out}/{tag}.ref.yuv', 'rb').read() == open(f'{out}/{tag}.my.yuv', 'rb').read()
    for suf in ('ref.yuv', 'my.yuv'):
        os.remove(f'{out}/{tag}.{suf}')
    return tag, 'OK' if (same_b and same_y) else f'MISMATCH bits={same_b} recon={same_y}'

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--jobs', type=int, default=8); ap.add_argument('--out', default='/tmp/pm')
    ap.add_argument('--sim', default=os.path.join(ROOT, 'tests', 'hostsim', 'hostsim')); ap.add_argument('--quick', action='store_true')
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    cases = []
    geos = [(192, 128), (208, 120), (144, 176), (352, 288)]
    for (w, h), qp, seed in itertools.product(geos, (22, 32, 40), (11, 12)):
        sigma = 2 if seed % 2 else 6
        frames = 4 if (w, h) == (352, 288) else 6
        cases.append((w, h, frames, seed, sigma, qp, ()))
    cases.append((192, 128, 14, 21, 3, 32, ()))            # crosses the HQperiod
    cases.append((192, 128, 6, 22, 3, 32, ('-max_num_ref', '2')))
    cases.append((192, 128, 6, 23, 3, 32, ('-enable_bipred', '0')))
    cases.append((192, 128, 6, 24, 3, 32, ('-cdef', '0', '-deblocking', '0')))
    cases.append((192, 128, 6, 25, 3, 32, ('-enable_tb_split', '0', '-enable_pb_split', '0')))
    cases.append((192, 128, 8, 26, 3, 30, ('-intra_period', '4')))
    cases.append((192, 128, 5, 27, 4, 46, ()))             # cdef_bits == 0 on P frames
    # the other operating points: hierarchical B (+ interpolated refs), encoder_speed 1/2, CLPF
    for cfgname in ('ra_high_efficiency.cfg', 'hdb16_high_efficiency.cfg', 'ldb_medium_complexity.cfg', 'ldb_low_complexity.cfg'):
        nfr = 18 if cfgname.startswith(('ra', 'hdb')) else 6
        for (w, h), qp, seed in (((192, 128), 32, 31), ((208, 120), 26, 32), ((320, 192), 38, 33)):
            cases.append((w, h, nfr, seed, 3, qp, (), cfgname))
    cases.append((192, 128, 6, 34, 3, 33, ('-clpf', '1'), 'ldb_high_efficiency.cfg'))     # CDEF and CLPF together
    cases.append((192, 128, 10, 35, 3, 30, ('-interp_ref', '0'), 'ra_high_efficiency.cfg'))
    ref_cfg = '/root/reference'
    if os.path.isdir(ref_cfg):  # the reference's own files for the remaining (non-qm) operating points
        for name in ('config_HDB_high_efficiency.txt', 'config_HDB_medium_complexity.txt', 'config_HDB_low_complexity.txt',
                     'config_RA_medium_complexity.txt', 'config_RA_low_complexity.txt', 'config_RA16_high_efficiency.txt',
                     'config_HDB16_medium_complexity.txt', 'config_HDB16_low_complexity.txt', 'config_RA16_low_complexity.txt'):
            cases.append((192, 128, 18, 36, 3, 32, (), os.path.join(ref_cfg, name)))
    if a.quick: cases = cases[:6]
    with ThreadPoolExecutor(a.jobs) as ex:
        bad = 0
        for tag, res in ex.map(lambda c: run_case(c, a.out, a.sim), cases):
            print(tag, res, flush=True)
            bad += res != 'OK'
    print('FAILED' if bad else 'ALL OK', bad)
    sys.exit(1 if bad else 0)
