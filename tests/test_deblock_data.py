"""deblock_data[] at the drop-in seam: after encode_frame_lbd/_hbd returns, encoder_info->deblock_data holds what
the reference's copy_deblock_data (enc/encode_block.c:1568-1613) would have left there - mode, cbp, size, tb_split, pb_part and
inter_pred of every 4x4 block.  Goldens: tests/golden/dd.npz, recorded from the reference by tests/golden/gen_dd.py."""
import json
import os
import numpy as np
import pytest
from util import ROOT, build_hostsim, HAVE_REFERENCE_TREE
from golden.gen_dd import record, CASES

G = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'streams.json')))
DD = np.load(os.path.join(ROOT, 'tests', 'golden', 'dd.npz'))
REF_HIPENC_DD = os.path.join(ROOT, 'oracle', '_ref', 'Thorenc_hip_dd')
REF_ENC_DD = os.path.join(ROOT, 'oracle', '_ref', 'Thorenc_dd')
FIELDS = 'mode cbp.y cbp.u cbp.v size tb_split pb_part mv0.x mv0.y mv1.x mv1.y ref_idx0 ref_idx1 bipred_flag'.split()


def check(name, order, dd):
    assert order.tolist() == DD[name + '/order'].tolist(), 'coding order'
    want = DD[name + '/dd'].astype(np.int32)
    assert dd.shape == want.shape
    for f in range(len(want)):
        for k, fn in enumerate(FIELDS):
            bad = np.nonzero(want[f, :, k] != dd[f, :, k])[0]
            assert len(bad) == 0, f'{name}: frame {order[f]} field {fn}: {len(bad)} cells differ, first at cell {bad[0]}: {dd[f, bad[0], k]} != {want[f, bad[0], k]}'


@pytest.mark.parametrize('name', ['192x128_n6_q32', '128x96_n9_q32_ra', '192x128_n5_q32_hdb16_gop4_10bit'])
def test_host_simulation_cells_equal_reference_deblock_data(name):
    """The engine's DbCells (host simulation of the device sources), converted by dd_fields() as the seam does, equal the
    reference's deblock_data[] after every frame - low delay, hierarchical B with interpolated references, 10 bit."""
    c = G[name]
    bits, rec, order, dd = record(build_hostsim(), c)
    check(name, order, dd)


@pytest.mark.skipif(not (HAVE_REFERENCE_TREE and os.path.exists(REF_ENC_DD)), reason='oracle/_ref/Thorenc_dd only exists in the build container')
def test_golden_deblock_data_is_what_the_live_reference_leaves():
    name = '128x96_n9_q32_ra'
    bits, rec, order, dd = record(REF_ENC_DD, G[name])
    check(name, order, dd)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_HIPENC_DD), reason='oracle/_ref/Thorenc_hip_dd not in the snapshot')
@pytest.mark.parametrize('name', CASES)
def test_dropin_seam_writes_deblock_data(name):
    """The reference front end linked against libthor_hip.so, encode_frame_lbd/_hbd wrapped by oracle/dd_shim.c: the array the
    library leaves in encoder_info->deblock_data after every frame equals the reference's, and stream + reconstruction still do."""
    from util import md5
    c = G[name]
    bits, rec, order, dd = record(REF_HIPENC_DD, c)
    assert md5(bits) == c['bit_md5'] and md5(rec) == c['rec_md5']
    check(name, order, dd)
