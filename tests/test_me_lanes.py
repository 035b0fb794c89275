"""The motion search with one lane per candidate (tk_me.h: me_cand_fullpel / me_cand8_subpel - what 64-lane teams run for PUs of up to 32x32
samples, i.e. what the MI355X runs; round 6: the full-pel passes on 16-bit samples too - 1008 more searches at bitdepth 10) against the generic search on the CPU: tests/hostsim/unit_me_lanes.cpp runs the product's motion_estimate over the
same sequences of searches with a team of 64 lanes (64 OS threads) and with a 1-lane team; vector and cost of every search must be equal (twelve PU
shapes x six variants - plain, reference "in the future", the other filter set, no staged window, one long candidate list with a large lambda, a small
lambda - 3024 searches; frame corners and edges, predictors on and far off the true motion, evolving candidate lists).  The CPU twin of the -DTK_ME_CROSSCHECK builds that make the same comparison inside the kernel
on the GPU.  (Sensitivity checked once by hand: a reversed tie-break or a rate term off by one quarter-pel in the lane-per-candidate code fails every search /
a third of the searches.)"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lane_per_candidate_search_equals_the_generic_search(tmp_path):
    exe = str(tmp_path / 'unit_me_lanes')
    subprocess.check_call(['g++', '-std=c++17', '-O2', '-fno-strict-aliasing', '-DTHOR_HOSTSIM', '-DTHOR_HOSTSIM_LANES=64', '-ffp-contract=off', '-pthread', '-o', exe,
                           os.path.join(ROOT, 'tests', 'hostsim', 'unit_me_lanes.cpp')])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0 and r.stdout.startswith('ok: 3024 searches (8-bit) + 1008 (16-bit)'), r.stdout + r.stderr
