#!/usr/bin/env python3
"""Event-driven model of the persistent superblock kernel's scheduler (thor_amd/csrc/tk_sched.h): W resident workgroups, S
streams of a rows x cols superblock grid with the codec's dependencies (left, up-right), synthetic superblock durations
calibrated to the measured distribution of a 4-reference 3840x2160 frame (profiles/archive/r03_sbtimes_4k_s96.log / _s144.log: mean
118-121 ms, median 116-119, p95 156-158, max 224-266; critical path per stream mean 8.1-8.3 s, max 9.1-9.4 s).
Compares queue disciplines: fifo (ready order, what the kernel has always done), lag (lowest anti-diagonal l + 2k first =
laggards first), stream / group priorities.   python3 scripts/sched_policy_model.py [seeds]"""
import heapq
import sys

import numpy as np


def make_durations(S, rows, cols, seed):
    rng = np.random.default_rng(seed)
    lvl = rng.normal(1.0, 0.045, size=(S, 1, 1))                     # stream level
    coarse = rng.normal(0.0, 1.0, size=(S, rows // 4 + 2, cols // 4 + 2))
    fld = np.repeat(np.repeat(coarse, 4, axis=1), 4, axis=2)[:, :rows, :cols]   # content varies smoothly over the frame
    noise = rng.normal(0.0, 1.0, size=(S, rows, cols))
    d = 119.0 * lvl * np.exp(0.13 * fld + 0.09 * noise - 0.012)
    heavy = rng.random(size=(S, rows, cols)) < 0.003
    return np.where(heavy, d * rng.uniform(1.3, 1.7, size=d.shape), d)


def critical_path(d):
    S, rows, cols = d.shape
    cp = np.zeros_like(d)
    for k in range(rows):
        for l in range(cols):
            dep = np.zeros(S)
            if l > 0:
                dep = np.maximum(dep, cp[:, k, l - 1])
            if k > 0:
                dep = np.maximum(dep, cp[:, k - 1, min(l + 1, cols - 1)])
            cp[:, k, l] = dep + d[:, k, l]
    return cp[:, -1, -1]


def simulate(d, W, policy, groups=1):
    S, rows, cols = d.shape
    need = [[(l > 0) + (k > 0) for l in range(cols)] for k in range(rows)]
    cnt = np.zeros((S, rows, cols), dtype=int)
    prio = {'fifo': lambda s, k, l: 0, 'lag': lambda s, k, l: l + 2 * k, 'stream': lambda s, k, l: s,
            'group': lambda s, k, l: s * groups // S}[policy]
    seq, ready, running = 0, [], []
    for s in range(S):
        heapq.heappush(ready, (prio(s, 0, 0), seq, s, 0, 0)); seq += 1
    free, t, done, total = W, 0.0, 0, S * rows * cols
    while done < total:
        while free > 0 and ready:
            _, _, s, k, l = heapq.heappop(ready)
            heapq.heappush(running, (t + d[s, k, l], s, k, l)); free -= 1
        t, s, k, l = heapq.heappop(running)
        free += 1; done += 1
        succ = []
        if l + 1 < cols:
            succ.append((k, l + 1))
        if k + 1 < rows:
            if l >= 1:
                succ.append((k + 1, l - 1))
            if l == cols - 1:
                succ.append((k + 1, l))
        for kk, ll in succ:
            cnt[s, kk, ll] += 1
            if cnt[s, kk, ll] == need[kk][ll]:
                heapq.heappush(ready, (prio(s, kk, ll), seq, s, kk, ll)); seq += 1
    return t


if __name__ == '__main__':
    seeds = [int(x) for x in sys.argv[1:]] or [7, 8, 9]
    rows, cols, W = 17, 30, 768
    print('| streams | SB ms mean / p95 / max | critical path s mean / max | work/768 s | fifo s (busy) | laggards first s (busy) | gain | stream priority | 2 groups |')
    print('|---|---|---|---|---|---|---|---|---|')
    for S in (96, 128, 144, 192):
        rowsout = []
        for seed in seeds:
            d = make_durations(S, rows, cols, seed)
            cp = critical_path(d)
            ideal = d.sum() / W
            r = {p: simulate(d, W, p, 2) for p in ('fifo', 'lag', 'stream', 'group')}
            rowsout.append((d.mean(), np.percentile(d, 95), d.max(), cp.mean() / 1e3, cp.max() / 1e3, ideal / 1e3, r['fifo'] / 1e3, ideal / r['fifo'],
                            r['lag'] / 1e3, ideal / r['lag'], r['fifo'] / r['lag'], ideal / r['stream'], ideal / r['group']))
        m = np.mean(np.array(rowsout), axis=0)
        print(f'| {S} | {m[0]:.0f} / {m[1]:.0f} / {m[2]:.0f} | {m[3]:.1f} / {m[4]:.1f} | {m[5]:.2f} | {m[6]:.2f} ({100 * m[7]:.0f} %) | {m[8]:.2f} ({100 * m[9]:.0f} %) | '
              f'+{100 * (m[10] - 1):.1f} % | {100 * m[11]:.0f} % busy | {100 * m[12]:.0f} % busy |')
