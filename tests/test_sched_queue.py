"""The ready-task queue of the persistent superblock kernel (thor_amd/csrc/tk_sched.h) run by OS threads: the same protocol
source the device compiles (atomics mapped to GCC builtins), grids from a single superblock to 4K's 17x30, more workers than
ready tasks and fewer.  The stress program checks that every task is handed out exactly once, never before its dependencies
have finished, and that data written by a dependency is visible to its successor."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, 'tests', 'hostsim', 'sched_stress')
SRC = EXE + '.cpp'
HDR = os.path.join(ROOT, 'thor_amd', 'csrc', 'tk_sched.h')


def _build():
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(['g++', '-std=c++17', '-O2', '-pthread', '-o', EXE, SRC])
    return EXE


@pytest.mark.parametrize('S,rows,cols,workers', [(1, 1, 1, 1), (7, 1, 1, 3), (3, 1, 7, 4), (3, 9, 1, 4), (5, 9, 15, 1), (64, 3, 3, 16),
                                                  (16, 17, 30, 12), (2, 17, 30, 24)])
def test_every_task_once_after_its_dependencies(S, rows, cols, workers):
    for seed in (1, 2, 3):
        r = subprocess.run([_build(), str(S), str(rows), str(cols), str(workers), str(seed)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr


@pytest.mark.parametrize('S,rows,cols,workers', [(2, 1, 1, 2), (4, 1, 7, 4), (4, 9, 1, 3), (6, 9, 15, 5), (16, 17, 30, 12), (3, 17, 30, 24), (8, 9, 15, 16)])
def test_staggered_half_frames_every_task_once_after_its_dependencies(S, rows, cols, workers):
    """The launches of Engine::encode_run (tk_encoder.h) for one frame of two stream groups half a frame apart: {group 0: first half},
    {group 0: second half, group 1: first half}, {group 1: second half} - ranges of anti-diagonals; dependencies that finished in an earlier
    launch are not counted, successors beyond the range are left to the next launch."""
    for seed in (1, 2):
        r = subprocess.run([_build(), str(S), str(rows), str(cols), str(workers), str(seed), '1'], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
