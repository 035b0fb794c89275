#!/usr/bin/env python3
"""Golden deblock_data[] arrays: what the REFERENCE's encode_frame leaves in encoder_info->deblock_data after every frame
(copy_deblock_data, enc/encode_block.c:1568-1613), recorded with oracle/_ref/Thorenc_dd (the reference front end with
encode_frame_lbd/_hbd wrapped by oracle/dd_shim.c; build container only: `make -C oracle ddenc`) on clips of
tests/golden/streams.json.  Stored as int16 [frames, cells, 14] per case in tests/golden/dd.npz, fields in the order of
dd_shim.c: mode, cbp.y, cbp.u, cbp.v, size, tb_split, pb_part, mv0.x, mv0.y, mv1.x, mv1.y, ref_idx0, ref_idx1, bipred_flag."""
import json, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from util import run_encoder, golden_clip, md5  # noqa: E402

CASES = ['192x128_n6_q32', '128x96_n9_q32_ra', '192x128_n5_q32_hdb16_gop4_10bit', '192x128_n6_q32_ldb_low', '208x120_n4_q38_ldb_medium_clpf']


def record(binary, c):
    """Runs a Thorenc-compatible front end that honours $THOR_DD_DUMP; returns (frame numbers in coding order, int32 [frames, cells, 14])."""
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, THOR_DD_DUMP=os.path.join(d, 'dd.bin'))
        bits, rec = run_encoder(binary, golden_clip(c['clip']), c['w'], c['h'], c['n'], c['qp'], c['extra'], env=env, cfg=c['cfg'])
        a = np.fromfile(env['THOR_DD_DUMP'], dtype=np.int32)
    cells = (c['w'] // 4) * (c['h'] // 4)
    a = a.reshape(-1, 3 + 14 * cells)
    assert (a[:, 0] == 0x44444444).all() and (a[:, 2] == cells).all()
    return bits, rec, a[:, 1].copy(), a[:, 3:].reshape(len(a), cells, 14)


def main():
    G = json.load(open(os.path.join(ROOT, 'tests/golden/streams.json')))
    out = {}
    for name in CASES:
        c = G[name]
        bits, rec, order, dd = record(os.path.join(ROOT, 'oracle/_ref/Thorenc_dd'), c)
        assert md5(bits) == c['bit_md5'] and md5(rec) == c['rec_md5'], name   # the wrapped front end is still the reference
        assert np.abs(dd).max() < 32768
        out[name + '/order'] = order.astype(np.int16)
        out[name + '/dd'] = dd.astype(np.int16)
        print(name, dd.shape, 'coding order', order.tolist())
    np.savez_compressed(os.path.join(ROOT, 'tests/golden/dd.npz'), **out)
    print('wrote dd.npz', os.path.getsize(os.path.join(ROOT, 'tests/golden/dd.npz')), 'bytes')


if __name__ == '__main__':
    main()
