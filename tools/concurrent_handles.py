"""Many handles in flight at once (enqueue all, then fetch all): every handle forks its inverse blocks to its own
three further streams, so dozens of k_trtri_block launches -- small persistent grids with stage barriers -- are in
flight beside hundreds of tile workgroups.  Every result must equal the same handle evaluated alone, bit for bit.
    python tools/concurrent_handles.py [handles=12] [rounds=6]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cugp_amd.gp as gp                                  # noqa: E402
from conftest import synth                                # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sizes = [1100, 1500, 1700, 2100, 2500, 3000, 900, 1300]
hs = []
for i in range(H):
    n = sizes[i % len(sizes)]
    X, y = synth(n, d=5, seed=100 + i)
    g = gp.Covsum(n, 5)
    g.set_data(X, y)
    hs.append(g)
hp = np.array([np.log(3.0), 0.0, np.log(0.1)])
alone = []
for g in hs:
    g.set_loghyperparam(hp)
    alone.append(g.loglik_grad())
t0 = time.time()
for r in range(rounds):
    for g in hs:
        g.set_loghyperparam(hp + 1e-3 * (r + 1))
        g.enqueue(True)
    for g in hs:
        g.fetch()
    for g in hs:
        g.set_loghyperparam(hp)
        g.enqueue(True)
    for i, g in enumerate(hs):
        ll, gr = g.fetch()
        assert ll == alone[i][0] and tuple(gr) == tuple(alone[i][1]), (i, ll, alone[i])
print("%d handles x %d rounds x 2 evaluations in flight together: results bit-equal to the evaluations alone (%.1f s)"
      % (H, rounds, time.time() - t0))
for g in hs:
    g.close()
