"""The oracle's C restatement (oracle/thor_oracle.c) is pinned against known-answer vectors recorded
from the REAL reference functions (tests/golden/kat.npz <- tests/golden/gen_kat.py)."""
import ctypes as C
import os
import numpy as np
import pytest
from util import GOLD, build_oracle_c, vp

K = np.load(os.path.join(GOLD, 'kat.npz'))
O = build_oracle_c()


def test_sad():
    plane = K['sad_plane']
    for i in range(7):
        org, cand, want = K[f'sad_org{i}'], K[f'sad_cand{i}'], K[f'sad_out{i}']
        h, w = org.shape
        got = [O.orc_sad(vp(org), w, C.c_void_p(int(plane.ctypes.data) + (12 + int(dy)) * 96 + 12 + int(dx)), 96, w, h) for dx, dy in cand]
        assert (np.array(got, dtype=np.uint32) == want).all()


def test_quote_mv_bits():
    got = [O.orc_quote_mv_bits(int(a), int(b)) for a, b in K['mvb_in']]
    assert (np.array(got) == K['mvb_out']).all()


def test_interp_luma():
    ref = np.ascontiguousarray(K['ip_ref'])
    pad, pw, ph = 16, 64, 48
    k = 0
    while f'ip_geo{k}' in K:
        w, h, bx, by, bip = [int(v) for v in K[f'ip_geo{k}']]
        for i, (mx, my) in enumerate(K[f'ip_mv{k}']):
            out = np.zeros((h, w), dtype=np.uint8)
            base = int(ref.ctypes.data) + (pad + by) * ref.shape[1] + pad + bx
            O.orc_interp_luma(vp(out), w, C.c_void_p(base), ref.shape[1], w, h, int(mx), int(my), 0, bip, pw, ph, bx, by)
            assert (out == K[f'ip_out{k}'][i]).all(), (k, i)
        k += 1
    assert k == 8


def test_tu_pipeline_and_coeff_bits():
    k = 0
    while f'tu_par{k}' in K:
        size, qp, ctype, fast = [int(v) for v in K[f'tu_par{k}']]
        org, pred = K[f'tu_org{k}'], K[f'tu_pred{k}']
        q = min(size, 16)
        for i in range(org.shape[0]):
            res = (org[i].astype(np.int16) - pred[i].astype(np.int16))
            fwd = np.zeros((q, q), dtype=np.int16)
            O.orc_fwd_transform(vp(np.ascontiguousarray(res)), vp(fwd), size, fast, 8)
            assert (fwd == K[f'tu_fwd{k}'][i]).all(), ('fwd', k, i)
            cq = np.zeros((q, q), dtype=np.int16); rec = np.zeros((size, size), dtype=np.uint8)
            cbp = O.orc_code_tu(vp(np.ascontiguousarray(org[i])), vp(np.ascontiguousarray(pred[i])), size, qp, ctype, fast, vp(cq), vp(rec))
            assert cbp == K[f'tu_cbp{k}'][i]
            assert (cq == K[f'tu_coefq{k}'][i]).all(), ('coefq', k, i)
            assert (rec == K[f'tu_rec{k}'][i]).all(), ('rec', k, i)
            if cbp:
                assert O.orc_coeff_bits(vp(cq), size, ctype) == K[f'tu_bits{k}'][i], ('bits', k, i)
        k += 1
    assert k == 24


K2 = np.load(os.path.join(GOLD, 'kat2.npz'))


def test_fast_subpel_approximations():
    """encoder_speed > 0: sad_calc_fasthalf / sad_calc_fastquarter (enc/encode_block.c:174, :286)."""
    plane = np.ascontiguousarray(K2['fs_plane'])
    k = 0
    while f'fs_geo{k}' in K2:
        w, h, bx, by = [int(v) for v in K2[f'fs_geo{k}']]
        org = np.ascontiguousarray(K2[f'fs_org{k}'])
        base = C.c_void_p(int(plane.ctypes.data) + by * 112 + bx)
        x, y = C.c_int(0), C.c_int(0)
        O.orc_fast_halfpel.restype = C.c_uint
        s = O.orc_fast_halfpel(vp(org), base, w, 112, w, h, C.byref(x), C.byref(y))
        assert [s, x.value, y.value] == [int(v) for v in K2[f'fs_half{k}']], ('half', k)
        for hx, hy, ws, wx, wy in K2[f'fs_quarter{k}']:
            x, y = C.c_int(int(hx)), C.c_int(int(hy))
            O.orc_fast_quarterpel.restype = C.c_uint
            s = O.orc_fast_quarterpel(vp(org), base, w, 112, w, h, C.byref(x), C.byref(y))
            assert [s, x.value, y.value] == [int(ws), int(wx), int(wy)], ('quarter', k, int(hx), int(hy))
        k += 1
    assert k == 36


def test_clpf_sample_and_block_statistics():
    """CLPF: clpf_sample (common/common_block.c:315) and detect_multi_clpf (enc/encode_block.c:2584)."""
    got = [O.orc_clpf_sample(*[int(t) for t in row], int(s), int(d)) for row, (s, d) in zip(K2['cs_in'], K2['cs_par'])]
    assert (np.array(got) == K2['cs_out']).all()
    org, rec = np.ascontiguousarray(K2['dm_org']), np.ascontiguousarray(K2['dm_rec'])
    H, W = org.shape
    for (x0, y0, dmp), want in zip(K2['dm_blocks'], K2['dm_sums']):
        s = (C.c_int * 4)(0, 0, 0, 0)
        O.orc_detect_multi_clpf(vp(rec), vp(org), int(x0), int(y0), W, H, W, W, s, 0, 8, int(dmp))
        assert list(s) == [int(v) for v in want], (int(x0), int(y0), int(dmp))


# ---- 16-bit samples: the same restatement against vectors recorded from the reference's _hbd functions (kat4.npz) ----
K4S = {10: np.load(os.path.join(GOLD, 'kat4.npz')), 12: np.load(os.path.join(GOLD, 'kat4_12.npz'))}   # round 6: bitdepth 12 too


@pytest.mark.parametrize('bd', [10, 12])
def test_sad_16bit(bd):
    K4 = K4S[bd]
    plane = np.ascontiguousarray(K4['sad_plane'])
    for i in range(6):
        org, cand, want = np.ascontiguousarray(K4[f'sad_org{i}']), K4[f'sad_cand{i}'], K4[f'sad_out{i}']
        h, w = org.shape
        got = [O.orc_sad16(vp(org), w, C.c_void_p(int(plane.ctypes.data) + 2 * ((12 + int(dy)) * 96 + 12 + int(dx))), 96, w, h) for dx, dy in cand]
        assert (np.array(got, dtype=np.uint32) == want).all()


@pytest.mark.parametrize('bd', [10, 12])
def test_interp_luma_16bit(bd):
    K4 = K4S[bd]
    ref = np.ascontiguousarray(K4['ip_ref'])
    pad, pw, ph = 16, 64, 48
    k = 0
    while f'ip_geo{k}' in K4:
        w, h, bx, by, bip = [int(v) for v in K4[f'ip_geo{k}']]
        for i, (mx, my) in enumerate(K4[f'ip_mv{k}']):
            out = np.zeros((h, w), dtype=np.uint16)
            base = int(ref.ctypes.data) + 2 * ((pad + by) * ref.shape[1] + pad + bx)
            O.orc_interp_luma16(vp(out), w, C.c_void_p(base), ref.shape[1], w, h, int(mx), int(my), 0, bip, pw, ph, bx, by, bd)
            assert (out == K4[f'ip_out{k}'][i]).all(), (k, i)
        k += 1
    assert k == 8


@pytest.mark.parametrize('bd', [10, 12])
def test_tu_pipeline_16bit(bd):
    K4 = K4S[bd]
    k = 0
    while f'tu_par{k}' in K4:
        size, qp, ctype, fast = [int(v) for v in K4[f'tu_par{k}']]
        org, pred = K4[f'tu_org{k}'], K4[f'tu_pred{k}']
        q = min(size, 16)
        for i in range(org.shape[0]):
            cq = np.zeros((q, q), dtype=np.int16); rec = np.zeros((size, size), dtype=np.uint16)
            cbp = O.orc_code_tu16(vp(np.ascontiguousarray(org[i])), vp(np.ascontiguousarray(pred[i])), size, qp, ctype, fast, vp(cq), vp(rec), bd)
            assert cbp == K4[f'tu_cbp{k}'][i]
            assert (cq == K4[f'tu_coefq{k}'][i]).all(), ('coefq', k, i)
            assert (rec == K4[f'tu_rec{k}'][i]).all(), ('rec', k, i)
        k += 1
    assert k == 24


def test_widesad():
    """widesad_calc / widesad_calc_simd (enc/encode_block.c:430-453, enc/enc_kernels.c:84-113): SAD at the offsets -3 -1 0 1 3, the smallest and its offset, the
    leftmost among equal ones - 8-bit vectors recorded through the file-static dispatcher with the scalar AND the SIMD path (they agree), 16-bit vectors from
    widesad_calc_simd_hbd at bitdepth 10 and 12 (tests/golden/gen_kat6.py -> kat6.npz; content with exact ties: flat, period-2 / period-4 columns)."""
    K6 = np.load(os.path.join(GOLD, 'kat6.npz'))
    x = C.c_int()
    planes = np.ascontiguousarray(K6['ws8_planes'])
    ties = 0
    for k in range(int(K6['ws8_n'])):
        org = np.ascontiguousarray(K6[f'ws8_org{k}'])
        pi, by, bx, w, h = (int(v) for v in K6[f'ws8_arg{k}'])
        want_s, want_x = (int(v) for v in K6[f'ws8_out{k}'])
        got = O.orc_widesad(vp(org), w, C.c_void_p(planes[pi].ctypes.data + by * 96 + bx), 96, w, h, C.byref(x))
        assert (got, x.value) == (want_s, want_x), (k, got, x.value, want_s, want_x)
        sads = [O.orc_sad(vp(org), w, C.c_void_p(planes[pi].ctypes.data + by * 96 + bx + o), 96, w, h) for o in (-3, -1, 0, 1, 3)]
        ties += sads.count(min(sads)) > 1
    assert ties >= 20, 'the vectors are meant to hold exact ties between offsets'
    for k in range(int(K6['ws16_n'])):
        org = np.ascontiguousarray(K6[f'ws16_org{k}'])
        bd, pi, by, bx = (int(v) for v in K6[f'ws16_arg{k}'])
        plane = np.ascontiguousarray(K6[f'ws16_planes_bd{bd}'][pi])
        want_s, want_x = (int(v) for v in K6[f'ws16_out{k}'])
        got = O.orc_widesad16(vp(org), 16, C.c_void_p(plane.ctypes.data + 2 * (by * 96 + bx)), 96, 16, 16, C.byref(x))
        assert (got, x.value) == (want_s, want_x), (k, bd, got, x.value, want_s, want_x)


def test_early_skip_sub_block_tests():
    """The two sub-block tests of the early-skip check (enc/encode_block.c:2146-2229; chroma in the form the encoder executes: calc_cbp_simd,
    enc/enc_kernels.c:828-907) - 288 vectors recorded from the reference's file-static functions around the decision boundary (tests/golden/gen_kat7.py)."""
    K7 = np.load(os.path.join(GOLD, 'kat7.npz'))
    O.orc_early_skip_sub.argtypes = O.orc_early_skip_subC.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
    org, pred, arg, want = (np.ascontiguousarray(K7[k]) for k in ('es_org', 'es_pred', 'es_arg', 'es_out'))
    for k in range(len(want)):
        chroma, size, qp, thr10 = (int(v) for v in arg[k])
        f = O.orc_early_skip_subC if chroma else O.orc_early_skip_sub
        got = f(vp(org[k]), 32, vp(pred[k]), 32, size, qp, thr10 / 10.0, 8)
        assert got == int(want[k]), (k, chroma, size, qp, thr10, got, int(want[k]))
    assert 0 < want.sum() < len(want)
