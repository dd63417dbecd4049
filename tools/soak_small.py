"""Soak of the paths round 6 touched, under uneven load: many small evaluations (the last block of k_trace takes the final
sums through a ticket hand-off, tuning key 12) on several handles IN FLIGHT TOGETHER, every result bit-equal to the same
handle's first answer at that point; then grouped experts behind the library exchange (a world of one).
    python tools/soak_small.py [rounds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cugp_amd.gp as gp                                  # noqa: E402
from conftest import synth                                # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sizes = [129, 300, 515, 777, 1100, 1500, 1537, 2049, 2500, 3000, 3500, 4096]
hp0 = np.array([np.log(3.0), 0.0, np.log(0.1)])
hs = []
for i, n in enumerate(sizes):
    X, y = synth(n, d=6, seed=n)
    g = gp.Covsum(n, 6)
    g.set_data(X, y)
    hs.append(g)
pts = [hp0 + 0.05 * k for k in range(4)]
first = {}
t0 = time.time()
count = 0
for r in range(rounds):
    p = pts[r % 4]
    for g in hs:
        g.set_loghyperparam(p)
        g.enqueue(True)
    for i, g in enumerate(hs):
        ll, gr = g.fetch()
        key = (i, r % 4)
        if key not in first:
            first[key] = (ll, tuple(gr))
        assert np.isfinite(ll) and (ll, tuple(gr)) == first[key], (sizes[i], r, ll, gr, first[key])
        count += 1
    if r % 50 == 0:
        print("round %d: %d evaluations, %.0f s" % (r, count, time.time() - t0), flush=True)
for g in hs:
    g.close()
X, y = synth(6 * 1500, d=6, seed=3)
b = gp.BCM.split(X, y, 6)
comm = gp.Comm(None, 0, 1, 0)
ref = None
for r in range(rounds):
    b.set_BCM_log_hyperparam(pts[r % 4])
    out = comm.loglik_grad_allgather(b, 6)
    if r < 4:
        first[("bcm", r)] = out.copy()
    assert np.array_equal(out, first[("bcm", r % 4)]), r
print("ok: %d single evaluations in flight together + %d grouped exchanges, all bit-stable, %.0f s" % (count, rounds, time.time() - t0))
