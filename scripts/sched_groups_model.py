#!/usr/bin/env python3
"""Model of staggered stream groups (companion of sched_policy_model.py): the streams of one GPU are split into `groups` groups
whose lock-step frames start a fraction of a frame apart; all groups share the 768 workgroup slots (work-conserving: a slot takes
any ready task), a group's next frame may start filt_ms (post-filters, host phases) after its previous one has finished.  This is
what two engines on two HIP streams would do if idle workgroups of one persistent launch left as soon as their launch had fewer
unclaimed tasks than live workgroups.  Not built - the numbers size the next step.
   python3 scripts/sched_groups_model.py"""
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sched_policy_model import make_durations  # noqa: E402

def run(S, groups, frames, W, filt_ms, policy, seed, offset_ms=0.0):
    """groups stream groups alternate: a group's frame n+1 may start filt_ms after its frame n has finished; all groups share
    the W workgroup slots (work-conserving: older frame first, then `policy` order)."""
    rows, cols = 17, 30
    per = S // groups
    d = {(g, f): make_durations(per, rows, cols, seed + 17 * g + f) for g in range(groups) for f in range(frames)}
    need = [[(l > 0) + (k > 0) for l in range(cols)] for k in range(rows)]
    cnt = {key: np.zeros((per, rows, cols), dtype=int) for key in d}
    left = {key: per * rows * cols for key in d}
    ready, running, events = [], [], []   # events: (time, 'start', g, f)
    seq = 0
    def prio(g, f, s, k, l):
        base = 0  # work conserving, no group priority
        return (l + 2 * k) if policy == 'lag' else 0
    def start_frame(g, f, t):
        nonlocal seq
        for s in range(per):
            heapq.heappush(ready, (prio(g, f, s, 0, 0), seq, g, f, s, 0, 0)); seq += 1
    for g in range(groups):
        heapq.heappush(events, (offset_ms * g / groups, g, 0))
    free, t, busy = W, 0.0, 0.0
    total = sum(left.values()); done = 0
    while done < total:
        # release frames whose start time has come
        while events and events[0][0] <= t:
            _, g, f = heapq.heappop(events); start_frame(g, f, t)
        while free > 0 and ready:
            _, _, g, f, s, k, l = heapq.heappop(ready)
            heapq.heappush(running, (t + d[(g, f)][s, k, l], g, f, s, k, l)); free -= 1
        # next event: task end or frame start
        tn = running[0][0] if running else float('inf')
        te = events[0][0] if events else float('inf')
        if te < tn and free > 0:
            t = te; continue
        t, g, f, s, k, l = heapq.heappop(running)
        free += 1; done += 1; busy += d[(g, f)][s, k, l]
        left[(g, f)] -= 1
        if left[(g, f)] == 0 and f + 1 < frames:
            heapq.heappush(events, (t + filt_ms, g, f + 1))
        succ = []
        if l + 1 < cols: succ.append((k, l + 1))
        if k + 1 < rows:
            if l >= 1: succ.append((k + 1, l - 1))
            if l == cols - 1: succ.append((k + 1, l))
        for kk, ll in succ:
            c = cnt[(g, f)]; c[s, kk, ll] += 1
            if c[s, kk, ll] == need[kk][ll]:
                heapq.heappush(ready, (prio(g, f, s, kk, ll), seq, g, f, s, kk, ll)); seq += 1
    return t, busy / W / t


if __name__ == '__main__':
    S = 128
    print('| groups | start offset | queue | stream-frames / s | slots busy (incl. the filter gaps) |')
    print('|---|---|---|---|---|')
    for groups, off in ((1, 0.0), (2, 6000.0), (4, 12000.0)):
        for pol in ('fifo', 'lag'):
            nf = 6 * groups
            t, eff = run(S, groups, nf, 768, 400.0, pol, 3, off)
            sf = (S // groups) * groups * nf
            print(f'| {groups} | {off / 1e3 / groups:.0f} s | {pol} | {sf / (t / 1e3):.2f} | {100 * eff:.0f} % |')
