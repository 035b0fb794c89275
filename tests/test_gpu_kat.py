"""Kernel-level entry points of the C ABI vs. the oracle's C restatement and the recorded KATs (-m gpu)."""
import ctypes as C
import os
import numpy as np
import pytest
from util import GOLD, build_oracle_c, vp

pytestmark = pytest.mark.gpu
K = np.load(os.path.join(GOLD, 'kat.npz'))


def test_sad_batch_matches_reference_kat():
    import thor_amd
    plane = K['sad_plane']
    for i in range(7):
        org, cand, want = K[f'sad_org{i}'], K[f'sad_cand{i}'], K[f'sad_out{i}']
        got = thor_amd.sad_batch(org, plane, 12, 12, cand)
        assert (got == want).all(), i


def test_sad_batch_random_vs_oracle_c():
    import thor_amd
    O = build_oracle_c()
    rng = np.random.default_rng(5)
    plane = rng.integers(0, 256, size=(200, 256), dtype=np.uint8)
    for (w, h) in ((4, 8), (8, 4), (16, 16), (64, 32), (128, 128)):
        org = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        cand = rng.integers(-30, 31, size=(400, 2)).astype(np.int32)
        got = thor_amd.sad_batch(org, plane, 32, 32, cand)
        want = [O.orc_sad(vp(org), w, C.c_void_p(int(plane.ctypes.data) + (32 + int(dy)) * 256 + 32 + int(dx)), 256, w, h) for dx, dy in cand]
        assert (got == np.array(want, dtype=np.uint32)).all()


def test_interp_luma_matches_reference_kat():
    import thor_amd
    ref = K['ip_ref']
    k = 0
    while f'ip_geo{k}' in K:
        w, h, bx, by, bip = [int(v) for v in K[f'ip_geo{k}']]
        got = thor_amd.interp_luma(ref, 16, 64, 48, bx, by, w, h, K[f'ip_mv{k}'], bip)
        assert (got == K[f'ip_out{k}']).all(), k
        k += 1


def test_code_tu_matches_reference_kat():
    import thor_amd
    k = 0
    while f'tu_par{k}' in K:
        size, qp, ctype, fast = [int(v) for v in K[f'tu_par{k}']]
        coefq, rec, cbp = thor_amd.code_tu_batch(K[f'tu_org{k}'], K[f'tu_pred{k}'], qp, ctype, fast)
        assert (cbp == K[f'tu_cbp{k}']).all(), k
        assert (coefq == K[f'tu_coefq{k}']).all(), k
        assert (rec == K[f'tu_rec{k}']).all(), k
        k += 1
    assert k == 24


def test_code_tu_random_edge_cases_vs_oracle_c():
    """Extreme residuals (+-255 everywhere, checkerboards, empty) - the wave-parallel quantiser and
    transforms against the serial C restatement."""
    import thor_amd
    O = build_oracle_c()
    rng = np.random.default_rng(9)
    for size in (4, 8, 16, 32, 64):
        n = 8
        org = rng.integers(0, 256, size=(n, size, size), dtype=np.uint8)
        pred = rng.integers(0, 256, size=(n, size, size), dtype=np.uint8)
        org[0] = 255; pred[0] = 0
        org[1] = 0; pred[1] = 255
        org[2] = pred[2]
        org[3] = (np.indices((size, size)).sum(0) % 2 * 255).astype(np.uint8); pred[3] = 255 - org[3]
        for qp, ctype in ((8, 0), (30, 2), (51, 1)):
            coefq, rec, cbp = thor_amd.code_tu_batch(org, pred, qp, ctype, 0)
            q = min(size, 16)
            for i in range(n):
                cq = np.zeros((q, q), dtype=np.int16); rc = np.zeros((size, size), dtype=np.uint8)
                c = O.orc_code_tu(vp(np.ascontiguousarray(org[i])), vp(np.ascontiguousarray(pred[i])), size, qp, ctype, 0, vp(cq), vp(rc))
                assert c == cbp[i] and (cq == coefq[i]).all() and (rc == rec[i]).all(), (size, qp, i)


def test_deblock_frame_matches_reference_kat():
    """thor_hip_deblock_frame (the four k_deblock passes) against deblock_frame_y_lbd / deblock_frame_uv_lbd of the reference
    (common/common_frame.c:47,354) on recorded random frames / block data (tests/golden/gen_kat3.py)."""
    import thor_amd
    K3 = np.load(os.path.join(GOLD, 'kat3.npz'))
    k = 0
    while f'db_par{k}' in K3:
        w, h, qp = [int(v) for v in K3[f'db_par{k}']]
        got = thor_amd.deblock_frame(K3[f'db_in{k}'], w, h, qp, K3[f'db_cells{k}'])
        assert (got == K3[f'db_out{k}']).all(), (k, int((got != K3[f'db_out{k}']).sum()))
        assert (got != K3[f'db_in{k}']).sum() > 100  # the filter did something
        k += 1
    assert k == 5


# ---- 16-bit samples: the _hbd entry points against vectors recorded from the reference's _hbd functions (kat4.npz, bitdepth 10)
K4 = np.load(os.path.join(GOLD, 'kat4.npz'))


def test_sad_batch_hbd_matches_reference_kat():
    import thor_amd
    plane = K4['sad_plane']
    for i in range(6):
        org, cand, want = K4[f'sad_org{i}'], K4[f'sad_cand{i}'], K4[f'sad_out{i}']
        got = thor_amd.sad_batch(org, plane, 12, 12, cand, bitdepth=10)
        assert (got == want).all(), i


def test_sad_batch_hbd_random_vs_oracle_c():
    import thor_amd
    O = build_oracle_c()
    rng = np.random.default_rng(6)
    plane = rng.integers(0, 1024, size=(200, 256), dtype=np.uint16)
    for (w, h) in ((4, 8), (8, 4), (16, 16), (64, 32), (128, 128)):
        org = rng.integers(0, 1024, size=(h, w), dtype=np.uint16)
        cand = rng.integers(-30, 31, size=(200, 2)).astype(np.int32)
        got = thor_amd.sad_batch(org, plane, 32, 32, cand, bitdepth=10)
        want = [O.orc_sad16(vp(org), w, C.c_void_p(int(plane.ctypes.data) + 2 * ((32 + int(dy)) * 256 + 32 + int(dx))), 256, w, h) for dx, dy in cand]
        assert (got == np.array(want, dtype=np.uint32)).all()


def test_interp_luma_hbd_matches_reference_kat():
    import thor_amd
    ref = K4['ip_ref']
    k = 0
    while f'ip_geo{k}' in K4:
        w, h, bx, by, bip = [int(v) for v in K4[f'ip_geo{k}']]
        got = thor_amd.interp_luma(ref, 16, 64, 48, bx, by, w, h, K4[f'ip_mv{k}'], bip, bitdepth=10)
        assert (got == K4[f'ip_out{k}']).all(), k
        k += 1
    assert k == 8


def test_code_tu_hbd_matches_reference_kat():
    import thor_amd
    k = 0
    while f'tu_par{k}' in K4:
        size, qp, ctype, fast = [int(v) for v in K4[f'tu_par{k}']]
        coefq, rec, cbp = thor_amd.code_tu_batch(K4[f'tu_org{k}'], K4[f'tu_pred{k}'], qp, ctype, fast, bitdepth=10)
        assert (cbp == K4[f'tu_cbp{k}']).all(), k
        assert (coefq == K4[f'tu_coefq{k}']).all(), k
        assert (rec == K4[f'tu_rec{k}']).all(), k
        k += 1
    assert k == 24


def test_deblock_frame_hbd_matches_reference_kat():
    import thor_amd
    k = 0
    while f'db_par{k}' in K4:
        w, h, qp = [int(v) for v in K4[f'db_par{k}']]
        got = thor_amd.deblock_frame(K4[f'db_in{k}'], w, h, qp, K4[f'db_cells{k}'], bitdepth=10)
        assert (got == K4[f'db_out{k}']).all(), (k, int((got != K4[f'db_out{k}']).sum()))
        assert (got != K4[f'db_in{k}']).sum() > 100
        k += 1
    assert k == 3
