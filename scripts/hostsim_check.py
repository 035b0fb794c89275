#!/usr/bin/env python3
"""hostsim_check.py BINARY [name-substring ...] - run a host-simulation build of the engine (tests/hostsim) over the small
reference goldens (tests/golden/streams.json) and report which differ.  Development aid; the test-suite runs the same check."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from util import golden_streams, golden_clip, run_encoder, md5

G = golden_streams()
binary = sys.argv[1]
filt = sys.argv[2:]
bad = 0
for name in sorted(G):
    if filt and not any(f in name for f in filt):
        continue
    c = G[name]
    t0 = time.time()
    bits, rec = run_encoder(binary, golden_clip(c['clip']), c['w'], c['h'], c['n'], c['qp'], c['extra'], cfg=c.get('cfg'))
    ok = md5(bits) == c['bit_md5'] and md5(rec) == c['rec_md5']
    bad += not ok
    print('%-40s %s  %.1fs' % (name, 'ok' if ok else 'DIFFERS', time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
