"""Full-size parity (-m gpu): the BASELINE.json configurations at their real geometry against golden stream / recon hashes
recorded from the reference encoder (tests/golden/gen_streams_big.py -> streams_big.json; the reference needs minutes to
an hour of CPU for these, so they are not run live).  Clips come from the seeded generator (thor_amd/synth.py).
  * 3840x2160 LDB_high_efficiency I+P (30x17 superblocks, 112-px last row, strides 4160/2080)  - config 4 unit / north star
  * 3840x2160 RA_high_efficiency qp 27, 9 frames = I, P + a full hierarchical-B sub-GOP with interpolated refs - config 3
  * 3840x2160 10-bit through the uint16 path (I P P) + a full 16-frame HDB16 sub-GOP at 416x240 10-bit   - config 5
  * 1920x1080 I+4P: all four references + bi-prediction over them                                         - config 2
  * 64 closed 1080p streams in one lock-step run, every stream against its own per-chunk reference run     - 8e / the bench regime
  * round 3, the frames the bench times: 3840x2160 I+5P (4 references + bi-prediction), 1080p 14 frames across HQperiod, and
    the 64-stream set with 6 frames per stream
  * round 4: 3840x2160 10-bit with a full 16-frame HDB16 sub-GOP (config 5 as specified), 3840x2160 I+5P on the hard (sigma 6) content
"""
import json
import os
import numpy as np
import pytest
from util import ROOT, GOLD, golden_clip, md5

pytestmark = pytest.mark.gpu
BIG = json.load(open(os.path.join(GOLD, 'streams_big.json')))


def _encode(c, clips, staggered=False):
    import thor_amd
    over = {}
    ex = list(c['extra'])
    while ex:
        k, v = ex.pop(0), ex.pop(0)
        over[k[1:]] = v
    p = thor_amd.load_config(os.path.join(ROOT, 'configs', c['cfg']), width=c['w'], height=c['h'], qp=c['qp'], f=30, **over)
    with thor_amd.Encoder(p, len(clips)) as enc:
        bits, recs = enc.encode_clips(clips, staggered=staggered)
        return bits, [b''.join(r.tobytes() for r in rs if r is not None) for rs in recs]


def _frames(c):
    raw = np.frombuffer(golden_clip(c['clip']), dtype=np.uint8)
    fsz = len(raw) // c['n']
    return [raw[f * fsz:(f + 1) * fsz] for f in range(c['n'])]


@pytest.mark.parametrize('name', [n for n in ('1080p_ldb_n5_q32', '4k_ldb_n2_q32', '4k_hdb16_10bit_n3_q32', 'hdb16_416x240_10bit_n17_q32',
                                              '4k_ra_n9_q27', '4k_ldb_n6_q32', '1080p_ldb_n14_q32', '4k_hdb16_10bit_n17_q32', '4k_ldb_sigma6_n6_q32',
                                              '4k_ra_n17_q27', '4k_ldb_12bit_n3_q32') if n in BIG])   # round 6: two RA sub-GOPs at 3840x2160, 12-bit samples at 3840x2160
def test_full_size_configuration_matches_reference_golden(name):
    c = BIG[name]
    bits, rec = _encode(c, [_frames(c)])
    assert len(bits[0]) == c['bit_bytes']
    assert md5(bits[0]) == c['bit_md5'], 'bitstream differs from the reference'
    assert md5(rec[0]) == c['rec_md5'], 'reconstruction differs from the reference'


def test_64_streams_1080p_six_frames_each_equals_its_reference_chunk():
    """The frames bench.py TIMES with the driver's flags: 64 different 1080p streams, I + 5 P - P 4 and P 5 search all four
    references and run the lock-step bi-prediction search over them - each stream hashed against its own reference run."""
    names = ['1080p_stream%02d_n6_q32' % s for s in range(64)]
    if names[0] not in BIG:
        pytest.skip('6-frame chunk goldens not generated')
    c0 = BIG[names[0]]
    bits, rec = _encode(c0, [_frames(BIG[n]) for n in names])
    bad = [n for i, n in enumerate(names) if md5(bits[i]) != BIG[n]['bit_md5'] or md5(rec[i]) != BIG[n]['rec_md5']]
    assert not bad, f'{len(bad)} of 64 streams differ from their reference chunk: {bad[:4]}'


def test_64_streams_1080p_each_equals_its_reference_chunk():
    """The regime bench.py times (many closed streams through the dependency FIFO in one launch per frame): 64 different
    1080p streams, I + P, each hashed against the reference run on exactly its frames."""
    names = ['1080p_stream%02d_n2_q32' % s for s in range(64)]
    c0 = BIG[names[0]]
    bits, rec = _encode(c0, [_frames(BIG[n]) for n in names])
    bad = [n for i, n in enumerate(names) if md5(bits[i]) != BIG[n]['bit_md5'] or md5(rec[i]) != BIG[n]['rec_md5']]
    assert not bad, f'{len(bad)} of 64 streams differ from their reference chunk: {bad[:4]}'


def test_64_streams_1080p_six_frames_staggered_groups_equal_their_reference_chunks():
    """thor_hip_encode_staged_run - what bench.py times since round 5: the 64 streams in two groups half a frame apart, every launch of the persistent
    kernel covering the second half of one group's frame and the first half of the other's (ranges of anti-diagonals of the 15 x 9 superblock
    grid, tk_sched.h); I + 5 P per stream, each stream hashed against its own reference run."""
    names = ['1080p_stream%02d_n6_q32' % s for s in range(64)]
    if names[0] not in BIG:
        pytest.skip('6-frame chunk goldens not generated')
    bits, rec = _encode(BIG[names[0]], [_frames(BIG[n]) for n in names], staggered=True)
    bad = [n for i, n in enumerate(names) if md5(bits[i]) != BIG[n]['bit_md5'] or md5(rec[i]) != BIG[n]['rec_md5']]
    assert not bad, f'{len(bad)} of 64 streams differ from their reference chunk: {bad[:4]}'


def test_nine_1080p_streams_27_frames_across_both_high_quality_frames_equal_their_reference_runs():
    """Round 5: the frame schedule of every frame bench.py times with the driver's flags (coded frames 5..24) and two more - LDB_high_efficiency past the
    second high-quality frame (coded frame 24), the long-term reference r1 = last HQ frame up to 12 frames back, the second lap of the 13-slot reference
    ring (enc/mainenc.c:455-500): the 27-frame 1080p clip and eight streams of bench.py's 1080p workload with 27 frames each, nine closed streams in
    one staggered run (thor_hip_encode_staged_run), each hashed against its own reference run."""
    names = ['1080p_ldb_n27_q32'] + ['1080p_stream%02d_n27_q32' % s for s in (0, 5, 10, 15, 21, 26, 31, 36)]
    names = [n for n in names if n in BIG]
    if len(names) < 2:
        pytest.skip('27-frame goldens not generated')
    bits, rec = _encode(BIG[names[0]], [_frames(BIG[n]) for n in names], staggered=True)
    bad = [n for i, n in enumerate(names) if md5(bits[i]) != BIG[n]['bit_md5'] or md5(rec[i]) != BIG[n]['rec_md5']]
    assert not bad, f'{len(bad)} of {len(names)} streams differ from their reference run: {bad}'
