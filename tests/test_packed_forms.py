"""Algebra of the packed / vectorised forms of the engine against the plain per-sample definitions (tests/hostsim/unit_forms.cpp, host
build of the engine headers): sub-pel samples as dot products - 8-bit `subk8_sample` / `subk8_strip`, 16-bit `subk16_sample` /
`subk16_strip` at 10 and 12 bits - against `luma_sample` for both filter sets, all 16 fractional positions, all 9 integer offsets, random
and extreme content; SSD as sum a^2 + sum b^2 - 2 sum ab modulo 2^32; truncating averages per dword; block copies in pieces;
2*org - pred in pieces of four samples."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packed_forms_equal_the_per_sample_definitions(tmp_path):
    exe = str(tmp_path / 'unit_forms')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-fno-strict-aliasing', '-DTHOR_HOSTSIM', '-ffp-contract=off', '-I', ROOT, '-include', os.path.join(ROOT, 'thor_amd', 'csrc', 'tk_tables.h'),
                           '-o', exe, os.path.join(ROOT, 'tests', 'hostsim', 'unit_forms.cpp')])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip() == 'ok', r.stderr
