"""Option handling of the C ABI (no GPU needed: every case is rejected before the library touches HIP).
The reference's table (enc/strings.c:287-356) is accepted name by name; what this path cannot encode bit-exactly is an
error, never silently ignored; thor_hip_open applies the reference's check_parameters() guards (enc/strings.c:470-555)."""
import ctypes as C
import os
import pytest
from util import ROOT

CFG = os.path.join(ROOT, 'configs', 'ldb_high_efficiency.cfg')


def _set(p, name, value):
    import thor_amd
    return thor_amd.lib().thor_hip_params_set(C.byref(p), name.encode(), str(value).encode())


def test_known_options_are_applied_and_unknown_or_unsupported_ones_rejected(tmp_path):
    import thor_amd
    p = thor_amd.load_config(CFG, width=640, height=360, qp=30, f=30)
    assert (p.width, p.height, p.qp, p.max_num_ref) == (640, 360, 30, 4)
    assert _set(p, '-cdef', 0) == 0 and p.cdef == 0
    assert _set(p, '-snrcalc', 0) == 0          # harmless reporting option of the reference: accepted
    assert _set(p, '-qmtx', 0) == 0             # unsupported feature at its default: fine
    assert _set(p, '-no_such_option', 1) == 1   # not in the reference's table
    for name, val in (('-qmtx', 1), ('-max_delta_qp', 2), ('-bitrate', 500), ('-sync', 1), ('-subsample', 444), ('-log2_sb_size', 6)):
        assert _set(p, name, val) == 2, name    # would change the bitstream: rejected
    assert _set(p, '-n', 10) == 3               # front-end option, not an encoder parameter
    with pytest.raises(ValueError):
        thor_amd.load_config(CFG, qmtx=1)
    bad = tmp_path / 'qm.cfg'
    bad.write_text('-max_num_ref 2 ; fine\n-qmtx 1 ; quantisation matrices\n')
    with pytest.raises(ValueError):
        thor_amd.load_config(str(bad))
    worse = tmp_path / 'typo.cfg'
    worse.write_text('-max_num_reff 2\n')
    with pytest.raises(ValueError):
        thor_amd.load_config(str(worse))


@pytest.mark.parametrize('over', [dict(HQperiod=0), dict(HQperiod=33), dict(num_reorder_pics=7, max_num_ref=1), dict(num_reorder_pics=3, intra_period=6),
                                  dict(num_reorder_pics=3, HQperiod=6), dict(width=100), dict(max_num_ref=5), dict(interp_ref=2), dict(num_reorder_pics=2),
                                  dict(bitdepth=10), dict(qp=60)])
def test_open_rejects_parameter_sets_the_reference_or_this_path_cannot_code(over):
    """check_parameters() guards + the limits of this path; rejected in thor_hip_open before any device call (returns NULL)."""
    import thor_amd
    kw = dict(width=640, height=360, qp=30, f=30)
    kw.update(over)
    p = thor_amd.load_config(CFG, **kw)
    with pytest.raises(RuntimeError):
        thor_amd.Encoder(p, 1)


def test_accessors_validate_their_arguments():
    import thor_amd
    L = thor_amd.lib()
    assert L.thor_hip_stream_bytes(None, 0) == 0
    assert L.thor_hip_stream_data(None, 0) is None
    assert L.thor_hip_stage_frame_device(None, 0, 0, None) != 0
    assert L.thor_hip_deblock_frame(None, 64, 64, 30, None) != 0
