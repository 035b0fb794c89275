"""The wavefront form of the CDEF search's distortion pass (thor_amd/csrc/tk_cdef.h: cdef_mse_block_wave - the form the device runs, one wavefront per 8x8
block: lanes = samples, then lanes = strengths, primary / secondary sums shared between the 64 strengths) against the plain per-(block, strength) form
built on cdef_filter_px (pinned to the reference's cdef_filter_block by the kat5 known answers): identical mse[] arrays on random frames with skipped blocks,
partial filter blocks at the frame edge, 8 / 10 / 12 bits and the three search speeds (tests/hostsim/unit_cdef.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cdef_wave_form_equals_the_plain_form(tmp_path):
    exe = str(tmp_path / 'unit_cdef')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-fno-strict-aliasing', '-DTHOR_HOSTSIM', '-ffp-contract=off', '-I', ROOT, '-include', os.path.join(ROOT, 'thor_amd', 'csrc', 'tk_tables.h'),
                           '-o', exe, os.path.join(ROOT, 'tests', 'hostsim', 'unit_cdef.cpp')])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip() == 'ok', r.stderr
