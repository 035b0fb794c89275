#!/usr/bin/env python3
"""Analyse a THOR_SBTIMES dump (per-superblock claim/start/end wall-clock stamps, 100 MHz) written by
libthor_hip.so: per frame makespan, busy time, dependency-wait share, duration spread."""
import struct, sys
import numpy as np
data = open(sys.argv[1], 'rb').read()
wgs = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
off = 0
while off < len(data):
    frame, S, nsb, cols = struct.unpack_from('4i', data, off); off += 16
    a = np.frombuffer(data, dtype=np.uint64, count=3 * S * nsb, offset=off).reshape(S, nsb, 3).astype(np.float64) / 1e5  # ms
    off += 24 * S * nsb
    m = a[:, :, 2] > 0   # superblocks of this launch (thor_hip_encode_staged_run launches cover half frames of the two stream groups)
    if not m.all():
        t0 = a[:, :, 0][m].min()
        dur = (a[:, :, 2] - a[:, :, 1])[m]
        mk = a[:, :, 2][m].max() - t0
        busy = dur.sum()
        per_stream = m.sum(axis=1)
        print(f'launch {frame}: {int(m.sum())} of {S * nsb} superblocks ({int((per_stream > 0).sum())} streams, {sorted(set(per_stream[per_stream > 0].tolist()))} per stream)  makespan {mk:9.1f} ms  '
              f'sum(SB) {busy / 1e3:9.2f} s  mean SB {dur.mean():7.1f} ms  ideal(busy/{wgs}) {busy / wgs:8.1f} ms  eff {busy / wgs / mk:5.2f}')
        continue
    t0 = a[:, :, 0].min()
    dur = a[:, :, 2] - a[:, :, 1]
    wait = a[:, :, 1] - a[:, :, 0]
    mk = a[:, :, 2].max() - t0
    busy = dur.sum()
    print(f'frame {frame}: S={S} nsb={nsb} makespan {mk:9.1f} ms  sum(SB) {busy / 1e3:9.2f} s  mean/med/p95/max SB {dur.mean():7.1f} {np.median(dur):7.1f} '
          f'{np.percentile(dur, 95):7.1f} {dur.max():7.1f} ms  wait-sum {wait.sum() / 1e3:8.2f} s  ideal(busy/{wgs}) {busy / wgs:8.1f} ms  eff {busy / wgs / mk:5.2f}')
    # critical path per stream (longest dependency chain of SB durations)
    rows = nsb // cols
    cp = np.zeros((S, rows, cols))
    d = dur.reshape(S, rows, cols)
    for k in range(rows):
        for l in range(cols):
            dep = np.zeros(S)
            if l > 0: dep = np.maximum(dep, cp[:, k, l - 1])
            if k > 0: dep = np.maximum(dep, cp[:, k - 1, min(l + 1, cols - 1)])
            cp[:, k, l] = dep + d[:, k, l]
    print(f'          critical path per stream: mean {cp[:, -1, -1].mean():9.1f} max {cp.max():9.1f} ms')
