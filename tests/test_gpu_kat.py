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
K4S = {10: np.load(os.path.join(GOLD, 'kat4.npz')), 12: np.load(os.path.join(GOLD, 'kat4_12.npz'))}   # round 6: the same families at bitdepth 12
K4 = K4S[10]


@pytest.mark.parametrize('bd', [10, 12])
def test_sad_batch_hbd_matches_reference_kat(bd):
    K4 = K4S[bd]
    import thor_amd
    plane = K4['sad_plane']
    for i in range(6):
        org, cand, want = K4[f'sad_org{i}'], K4[f'sad_cand{i}'], K4[f'sad_out{i}']
        got = thor_amd.sad_batch(org, plane, 12, 12, cand, bitdepth=bd)
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


@pytest.mark.parametrize('bd', [10, 12])
def test_interp_luma_hbd_matches_reference_kat(bd):
    K4 = K4S[bd]
    import thor_amd
    ref = K4['ip_ref']
    k = 0
    while f'ip_geo{k}' in K4:
        w, h, bx, by, bip = [int(v) for v in K4[f'ip_geo{k}']]
        got = thor_amd.interp_luma(ref, 16, 64, 48, bx, by, w, h, K4[f'ip_mv{k}'], bip, bitdepth=bd)
        assert (got == K4[f'ip_out{k}']).all(), k
        k += 1
    assert k == 8


@pytest.mark.parametrize('bd', [10, 12])
def test_code_tu_hbd_matches_reference_kat(bd):
    K4 = K4S[bd]
    import thor_amd
    k = 0
    while f'tu_par{k}' in K4:
        size, qp, ctype, fast = [int(v) for v in K4[f'tu_par{k}']]
        coefq, rec, cbp = thor_amd.code_tu_batch(K4[f'tu_org{k}'], K4[f'tu_pred{k}'], qp, ctype, fast, bitdepth=bd)
        assert (cbp == K4[f'tu_cbp{k}']).all(), k
        assert (coefq == K4[f'tu_coefq{k}']).all(), k
        assert (rec == K4[f'tu_rec{k}']).all(), k
        k += 1
    assert k == 24


@pytest.mark.parametrize('bd', [10, 12])
def test_deblock_frame_hbd_matches_reference_kat(bd):
    K4 = K4S[bd]
    import thor_amd
    k = 0
    while f'db_par{k}' in K4:
        w, h, qp = [int(v) for v in K4[f'db_par{k}']]
        got = thor_amd.deblock_frame(K4[f'db_in{k}'], w, h, qp, K4[f'db_cells{k}'], bitdepth=bd)
        assert (got == K4[f'db_out{k}']).all(), (k, int((got != K4[f'db_out{k}']).sum()))
        assert (got != K4[f'db_in{k}']).sum() > 100
        k += 1
    assert k == 3


# ---- round 6: the sample kernels that were only covered by whole-stream hashes, against vectors recorded from the reference functions
# (tests/golden/gen_kat5.py -> kat5.npz; bitdepth 8 = the _lbd instances, 10 / 12 = _hbd) ------------------------------------------------
K5 = np.load(os.path.join(GOLD, 'kat5.npz'))
BDS = [8, 10, 12]


@pytest.mark.parametrize('bd', BDS)
def test_intra_prediction_matches_reference_kat(bd):
    """make_top_and_left + get_intra_prediction (common/intra_prediction.c:57-183, :403-428): every mode, blocks of 4..32 at the frame corner / top row /
    left column / interior with and without up-right / down-left samples, and the four transform units of split blocks (edges from the block-local
    reconstruction)."""
    import thor_amd
    plane = K5[f'in{bd}_plane']
    k = 0
    while f'in{bd}_geo{k}' in K5:
        size, tb = [int(v) for v in K5[f'in{bd}_geo{k}']]
        got = thor_amd.binding.kat_intra(plane, size, K5[f'in{bd}_par{k}'], bd, K5[f'in{bd}_rb{k}'] if tb else None)
        want = K5[f'in{bd}_out{k}']
        bad = [i for i in range(len(want)) if (got[i] != want[i]).any()]
        assert not bad, (k, size, tb, [K5[f'in{bd}_par{k}'][i].tolist() for i in bad[:4]])
        k += 1
    assert k == 7


@pytest.mark.parametrize('bd', BDS)
def test_inter_prediction_yuv_matches_reference_kat(bd):
    """get_inter_prediction_yuv (common/inter_prediction.c:185-226): clip_mv, quarter-pel luma, eighth-pel chroma (the `sic` clamp of :78), one PU and
    four quadrant PUs, vectors far outside the frame."""
    import thor_amd
    W, H = [int(v) for v in K5[f'ip{bd}_geo']]
    k = 0
    while f'ip{bd}_size{k}' in K5:
        size = int(K5[f'ip{bd}_size{k}'][0])
        got = thor_amd.binding.kat_inter_yuv(K5[f'ip{bd}_yuv'], W, H, size, K5[f'ip{bd}_par{k}'], K5[f'ip{bd}_mv{k}'], bd)
        want = K5[f'ip{bd}_out{k}']
        bad = [i for i in range(len(want)) if (got[i] != want[i]).any()]
        assert not bad, (k, size, [(K5[f'ip{bd}_par{k}'][i].tolist(), K5[f'ip{bd}_mv{k}'][i].tolist()) for i in bad[:3]])
        k += 1
    assert k == 4


@pytest.mark.parametrize('bd', BDS)
def test_average_blocks_matches_reference_kat(bd):
    import thor_amd
    for k, size in enumerate((8, 32)):
        got = thor_amd.binding.kat_average(K5[f'av{bd}_a{k}'], K5[f'av{bd}_b{k}'], size, bd)
        assert (got == K5[f'av{bd}_out{k}']).all(), k


@pytest.mark.parametrize('bd', BDS)
def test_chroma_from_luma_matches_reference_kat(bd):
    """improve_uv_prediction (common/common_block.c:347-428): items with a good luma prediction (untouched), correlated and uncorrelated chroma."""
    import thor_amd
    k = 0
    while f'cf{bd}_n{k}' in K5:
        n = int(K5[f'cf{bd}_n{k}'][0])
        got = thor_amd.binding.kat_cfl(K5[f'cf{bd}_y{k}'], K5[f'cf{bd}_uv{k}'], K5[f'cf{bd}_ry{k}'], n, bd)
        assert (got == K5[f'cf{bd}_out{k}']).all(), (k, n)
        assert (K5[f'cf{bd}_out{k}'] != K5[f'cf{bd}_uv{k}']).any()
        k += 1
    assert k == 4


@pytest.mark.parametrize('bd', BDS)
def test_cdef_direction_and_filter_match_reference_kat(bd):
    """cdef_find_dir (common/common_block.c:94-162) and cdef_filter_block (:224-279; recorded from the SIMD kernel the binary executes): 8x8 luma and 4x4
    chroma blocks, frame corners (CDEF_VERY_LARGE taps), every direction, primary / secondary strengths and dampings."""
    import thor_amd
    d, v = thor_amd.binding.kat_cdef_dir(K5[f'cd{bd}_blocks'], bd)
    assert (d == K5[f'cd{bd}_dir']).all() and (v == K5[f'cd{bd}_var']).all()
    assert len(set(d.tolist())) == 8
    for k, bsize in enumerate((8, 4)):
        got = thor_amd.binding.kat_cdef_filter(K5[f'cd{bd}_plane'], bsize, K5[f'cd{bd}_fpar{k}'], bd)
        want = K5[f'cd{bd}_fout{k}']
        bad = [i for i in range(len(want)) if (got[i] != want[i]).any()]
        assert not bad, (bsize, [K5[f'cd{bd}_fpar{k}'][i].tolist() for i in bad[:4]])


@pytest.mark.parametrize('bd', BDS)
def test_clpf_statistics_and_filter_match_reference_kat(bd):
    """CLPF: the per-block squared errors of detect_multi_clpf and the frame filtered by clpf_block under clpf_frame's skip / filter-block / boundary rules."""
    import thor_amd
    W, H, qp, fb_log2, s0, s1, s2 = [int(v) for v in K5[f'cl{bd}_par']]
    stats, out = thor_amd.binding.kat_clpf(K5[f'cl{bd}_rec'], K5[f'cl{bd}_org'], W, H, qp, K5[f'cl{bd}_cells'], [s0, s1, s2], fb_log2, K5[f'cl{bd}_fb_on'], bd)
    assert (stats == K5[f'cl{bd}_stats']).all(), np.argwhere(stats != K5[f'cl{bd}_stats'])[:4].tolist()
    assert (out == K5[f'cl{bd}_out']).all(), int((out != K5[f'cl{bd}_out']).sum())
    assert (out != K5[f'cl{bd}_rec']).sum() > 500


@pytest.mark.parametrize('bd', [8, 10])
def test_interpolate_frames_matches_reference_kat(bd):
    """interpolate_frames(new, ref0, ref1, 2, 1) (common/temporal_interp.c:909) on a panned 192x128 pair: the device pyramid / block motion estimation /
    merge / motion-compensated average against the reference's frame, luma and chroma."""
    import thor_amd
    W, H = [int(v) for v in K5[f'it{bd}_geo']]
    got = thor_amd.binding.kat_interpolate(K5[f'it{bd}_a'], K5[f'it{bd}_b'], W, H, bd)
    want = K5[f'it{bd}_out']
    assert (got == want).all(), int((got != want).sum())
