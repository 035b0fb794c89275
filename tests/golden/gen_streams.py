#!/usr/bin/env python3
"""Record golden stream / reconstruction hashes from the reference encoder (oracle/_ref/Thorenc) for
the small committed clips.  Run in the build container after `make -C oracle`; output
tests/golden/streams.json is committed and is what the GPU tests / smoke() compare with."""
import gzip, hashlib, json, os, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G = os.path.join(ROOT, 'tests', 'golden')
CASES = {
    '192x128_n3_q32': ('clip_192x128_6.yuv.gz', 192, 128, 3, 32, []),
    '192x128_n6_q32': ('clip_192x128_6.yuv.gz', 192, 128, 6, 32, []),
    '192x128_n6_q24': ('clip_192x128_6.yuv.gz', 192, 128, 6, 24, []),
    '192x128_n4_q44': ('clip_192x128_6.yuv.gz', 192, 128, 4, 44, []),
    '192x128_n3_q32_skip3': ('clip_192x128_6.yuv.gz', 192, 128, 3, 32, ['-skip', '3']),
    '208x120_n4_q32': ('clip_208x120_4.yuv.gz', 208, 120, 4, 32, []),
    '208x120_n4_q36_nocdef': ('clip_208x120_4.yuv.gz', 208, 120, 4, 36, ['-cdef', '0']),
    # 10-bit samples (uint16 LE), the reference's _hbd code path
    '192x128_n4_q32_10bit': ('clip10_192x128_5.yuv.gz', 192, 128, 4, 32, ['-bitdepth', '10', '-input_bitdepth', '10']),
    '192x128_n3_q40_10bit_nocdef': ('clip10_192x128_5.yuv.gz', 192, 128, 3, 40, ['-bitdepth', '10', '-input_bitdepth', '10', '-cdef', '0']),
    # hierarchical-B with temporally interpolated references (RA / HDB16 operating points)
    '128x96_n9_q32_ra': ('clip_128x96_9.yuv.gz', 128, 96, 9, 32, [], 'ra_high_efficiency.cfg'),
    '192x128_n6_q30_ra_gop4': ('clip_192x128_6.yuv.gz', 192, 128, 6, 30, ['-num_reorder_pics', '3'], 'ra_high_efficiency.cfg'),
    '192x128_n6_q36_ra_gop4_nointerp': ('clip_192x128_6.yuv.gz', 192, 128, 6, 36, ['-num_reorder_pics', '3', '-interp_ref', '0'], 'ra_high_efficiency.cfg'),
    '192x128_n5_q32_hdb16_gop4_10bit': ('clip10_192x128_5.yuv.gz', 192, 128, 5, 32,
                                        ['-num_reorder_pics', '3', '-bitdepth', '10', '-input_bitdepth', '10'], 'hdb16_high_efficiency.cfg'),
    # low / medium complexity operating points: encoder_speed 2 / 1, CLPF
    '192x128_n6_q32_ldb_low': ('clip_192x128_6.yuv.gz', 192, 128, 6, 32, [], 'ldb_low_complexity.cfg'),
    '208x120_n4_q30_ldb_medium': ('clip_208x120_4.yuv.gz', 208, 120, 4, 30, [], 'ldb_medium_complexity.cfg'),
    '208x120_n4_q38_ldb_medium_clpf': ('clip_208x120_4.yuv.gz', 208, 120, 4, 38, ['-clpf', '1'], 'ldb_medium_complexity.cfg'),
    '192x128_n5_q34_ldb_low_10bit': ('clip10_192x128_5.yuv.gz', 192, 128, 5, 34, ['-bitdepth', '10', '-input_bitdepth', '10'], 'ldb_low_complexity.cfg'),
    # BASELINE config 1 exactly: 352x288, 30 frames, seed 1, LDB_low_complexity, qp 32 (clip generated, not committed)
    'cfg1_352x288_n30_q32_ldb_low': ('gen:352,288,30,1,2.0', 352, 288, 30, 32, [], 'ldb_low_complexity.cfg'),
    # round 5: the frame schedule bench.py's timed frames depend on - LDB_high_efficiency past the second high-quality frame (coded frame 24) and
    # into the second lap of the 13-slot reference ring (enc/mainenc.c:455-500: long-term reference r1 = last HQ frame)
    '192x128_n27_q32_ldb': ('gen:192,128,27,7,2.0', 192, 128, 27, 32, []),
    '208x120_n26_q24_ldb': ('gen:208,120,26,8,3.0', 208, 120, 26, 24, []),
    # round 6: 12-bit samples (enc/strings.c:552-554 allows 8 / 10 / 12) - the claim "10/12-bit" of DESIGN 0 had no evidence before
    '192x128_n4_q32_12bit': ('gen:192,128,4,11,2.0,12', 192, 128, 4, 32, ['-bitdepth', '12', '-input_bitdepth', '12']),
    '192x128_n5_q30_hdb16_gop4_12bit': ('gen:192,128,5,12,2.5,12', 192, 128, 5, 30,
                                        ['-num_reorder_pics', '3', '-bitdepth', '12', '-input_bitdepth', '12'], 'hdb16_high_efficiency.cfg'),
    '208x120_n4_q36_ldb_medium_clpf_12bit': ('gen:208,120,4,13,2.0,12', 208, 120, 4, 36, ['-clpf', '1', '-bitdepth', '12', '-input_bitdepth', '12'],
                                             'ldb_medium_complexity.cfg'),
}
out = {}
with tempfile.TemporaryDirectory() as d:
    for name, case in CASES.items():
        clip, w, h, n, qp, extra = case[:6]
        cfg = case[6] if len(case) > 6 else 'ldb_high_efficiency.cfg'
        import sys
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from util import golden_clip
        raw = golden_clip(clip)
        open(os.path.join(d, 'in.yuv'), 'wb').write(raw)
        log = subprocess.run([os.path.join(ROOT, 'oracle/_ref/Thorenc'), '-cf', os.path.join(ROOT, 'configs', cfg),
                              '-if', os.path.join(d, 'in.yuv'), '-width', str(w), '-height', str(h), '-qp', str(qp), '-n', str(n),
                              '-f', '30', '-of', os.path.join(d, 'o.bit'), '-rf', os.path.join(d, 'o.yuv')] + extra,
                             check=True, capture_output=True, text=True).stdout
        frames = [l.split()[:4] for l in log.splitlines() if len(l.split()) > 4 and l.split()[1] in 'IPB']
        out[name] = {'clip': clip, 'cfg': cfg, 'w': w, 'h': h, 'n': n, 'qp': qp, 'extra': extra,
                     'bit_md5': hashlib.md5(open(os.path.join(d, 'o.bit'), 'rb').read()).hexdigest(),
                     'rec_md5': hashlib.md5(open(os.path.join(d, 'o.yuv'), 'rb').read()).hexdigest(),
                     'bit_bytes': os.path.getsize(os.path.join(d, 'o.bit')), 'frames': frames}
json.dump(out, open(os.path.join(G, 'streams.json'), 'w'), indent=1)
print('wrote', len(out), 'cases')
