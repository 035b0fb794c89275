"""The motion search with one lane per candidate (tk_me.h: me_cand8_fullpel / me_cand8_subpel - what 64-lane teams run for 8-bit PUs of up to 32x32
samples, i.e. what the MI355X runs) against the generic search on the CPU: tests/hostsim/unit_me_lanes.cpp runs the product's motion_estimate over the
same sequence of searches with a team of 64 lanes (64 OS threads) and with a 1-lane team; vector and cost of every search must be equal.  The CPU twin of
the -DTK_ME_CROSSCHECK builds that make the same comparison inside the kernel on the GPU.  A slice of the PU shapes per test (64 threads per team are
slow); THOR_ME_LANES_ALL=1 runs every shape."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_EXE = {}


def _exe(tmp_path_factory):
    if 'p' not in _EXE:
        out = str(tmp_path_factory.mktemp('me_lanes') / 'unit_me_lanes')
        subprocess.check_call(['g++', '-std=c++17', '-O2', '-DTHOR_HOSTSIM', '-DTHOR_HOSTSIM_LANES=64', '-ffp-contract=off', '-pthread', '-o', out,
                               os.path.join(ROOT, 'tests', 'hostsim', 'unit_me_lanes.cpp')])
        _EXE['p'] = out
    return _EXE['p']


# index into unit_me_lanes.cpp's list of PU shapes: 0 4x4, 1 8x8, 2 8x4, 3 4x8, 4 16x16, 5 16x8, 6 8x16, 7 32x32, 8 32x16, 9 16x32, 10 32x8, 11 8x32
_SLICE = list(range(12)) if os.environ.get('THOR_ME_LANES_ALL') else [0, 5, 9]


@pytest.mark.parametrize('case', _SLICE)
def test_lane_per_candidate_search_equals_the_generic_search(case, tmp_path_factory):
    r = subprocess.run([_exe(tmp_path_factory), str(case), '1'], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and r.stdout.startswith('ok:'), r.stdout + r.stderr
