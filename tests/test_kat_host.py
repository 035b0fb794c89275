"""Round 6: the known answers of tests/golden/kat5.npz (recorded from the reference functions, tests/golden/gen_kat5.py) against the 1-lane HOST build of the
device functions (tests/hostsim/kat_host.cpp: make_edges + pred_intra, the two CLPF passes, cdef_find_dir, cdef_filter_px) - the CPU twin of the
device known-answer tests (tests/test_gpu_kat.py), bitdepth 8 / 10 / 12."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K5 = np.load(os.path.join(ROOT, 'tests', 'golden', 'kat5.npz'))
P = lambda a: a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope='module')
def L(tmp_path_factory):
    so = str(tmp_path_factory.mktemp('kat') / 'kat_host.so')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-fno-strict-aliasing', '-DTHOR_HOSTSIM', '-ffp-contract=off', '-shared', '-fPIC', '-o', so,
                           os.path.join(ROOT, 'tests', 'hostsim', 'kat_host.cpp')])
    return C.CDLL(so)


@pytest.mark.parametrize('bd', [8, 10, 12])
def test_intra_host_build_matches_reference_kat(L, bd):
    plane = np.ascontiguousarray(K5[f'in{bd}_plane'])
    k = 0
    while f'in{bd}_geo{k}' in K5.files:
        size, tb = [int(v) for v in K5[f'in{bd}_geo{k}']]
        par = np.ascontiguousarray(K5[f'in{bd}_par{k}'])
        want = K5[f'in{bd}_out{k}']
        rb = np.ascontiguousarray(K5[f'in{bd}_rb{k}']) if tb else None
        out = np.zeros_like(want)
        L.h_intra(P(plane), plane.shape[1], bd, size, tb, len(par), P(par), P(rb) if tb else None, P(out))
        assert (out == want).all(), (k, size, tb)
        k += 1
    assert k == 7


@pytest.mark.parametrize('bd', [8, 10, 12])
def test_clpf_and_cdef_host_build_match_reference_kat(L, bd):
    W, H, qp, fbl, s0, s1, s2 = [int(v) for v in K5[f'cl{bd}_par']]
    rec, org = np.ascontiguousarray(K5[f'cl{bd}_rec']), np.ascontiguousarray(K5[f'cl{bd}_org'])
    cells, fb = np.ascontiguousarray(K5[f'cl{bd}_cells']), np.ascontiguousarray(K5[f'cl{bd}_fb_on'])
    nblk = (W // 8) * (H // 8) + 2 * (W // 16) * (H // 16)
    stats = np.zeros((nblk, 4), dtype=np.uint32)
    out = np.zeros_like(rec)
    st = np.array([s0, s1, s2], dtype=np.int32)
    L.h_clpf(P(rec), P(org), W, H, bd, qp, P(cells), P(st), fbl, P(fb), P(stats), P(out))
    assert (stats == K5[f'cl{bd}_stats']).all() and (out == K5[f'cl{bd}_out']).all() and (out != rec).sum() > 500
    blocks = np.ascontiguousarray(K5[f'cd{bd}_blocks'])
    d = np.zeros(len(blocks), dtype=np.int32)
    v = np.zeros(len(blocks), dtype=np.int32)
    L.h_cdef_dir(P(blocks), bd, len(blocks), P(d), P(v))
    assert (d == K5[f'cd{bd}_dir']).all() and (v == K5[f'cd{bd}_var']).all()
    plane = np.ascontiguousarray(K5[f'cd{bd}_plane'])
    for k, bsize in enumerate((8, 4)):
        par = np.ascontiguousarray(K5[f'cd{bd}_fpar{k}'])
        want = K5[f'cd{bd}_fout{k}']
        got = np.zeros_like(want)
        L.h_cdef_filter(P(plane), plane.shape[1], plane.shape[0], plane.shape[1], bd, bsize, len(par), P(par), P(got))
        assert (got == want).all(), bsize


def test_early_skip_sub_block_tests_host_build_match_reference_kat(tmp_path):
    """early_skip_sub / early_skip_subC (tk_block.h; the luma and the chroma sub-block test of check_early_skip) in the 1-lane host build against the 288 known
    answers recorded from the reference's check_early_skip_sub_block / _sub_blockC with use_simd = 1 (tests/golden/gen_kat7.py) - luma 8 / 16 / 32, chroma 4 / 8 / 16
    (incl. the 4-wide column-pair form of calc_cbp_simd), three qp, two thresholds, residuals around the decision boundary."""
    so = str(tmp_path / 'kat_host_es.so')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-fno-strict-aliasing', '-DTHOR_HOSTSIM', '-ffp-contract=off', '-shared', '-fPIC', '-o', so,
                           os.path.join(ROOT, 'tests', 'hostsim', 'kat_host_es.cpp')])
    H = C.CDLL(so)
    H.h_early_skip_sub.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
    K7 = np.load(os.path.join(ROOT, 'tests', 'golden', 'kat7.npz'))
    org, pred, arg, want = (np.ascontiguousarray(K7[k]) for k in ('es_org', 'es_pred', 'es_arg', 'es_out'))
    for k in range(len(want)):
        chroma, size, qp, thr10 = (int(v) for v in arg[k])
        got = H.h_early_skip_sub(chroma, P(org[k]), 32, P(pred[k]), 32, size, qp, thr10 / 10.0, 8)
        assert got == int(want[k]), (k, chroma, size, qp, thr10, got, int(want[k]))
