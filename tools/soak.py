"""Soak: many evaluations over a spread of sizes and hyper-parameters on one handle per size, every result compared
with a second evaluation at the same point (bit-equal) and checked finite; CG runs at small sizes.  A few minutes.
    python tools/soak.py [rounds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cugp_amd.gp as gp                                  # noqa: E402
from conftest import synth                                # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
sizes = [128, 129, 777, 1500, 1537, 2049, 3000, 4097, 6000, 8192]
rng = np.random.default_rng(7)
t0 = time.time()
hs = {}
for n in sizes:
    X, y = synth(n, d=6, seed=n)
    g = gp.Covsum(n, 6)
    g.set_data(X, y)
    hs[n] = g
count = 0
for r in range(rounds):
    for n, g in hs.items():
        hp = np.array([np.log(3.0), 0.0, np.log(0.1)]) + 0.2 * rng.standard_normal(3)
        g.set_loghyperparam(hp)
        a = g.loglik_grad()
        g.set_loghyperparam(hp + 0.5)
        g.loglik_grad()
        g.set_loghyperparam(hp)
        b = g.loglik_grad()
        assert np.isfinite(a[0]) and np.all(np.isfinite(a[1])), (n, hp, a)
        assert a[0] == b[0] and tuple(a[1]) == tuple(b[1]), (n, hp, a, b)
        count += 3
    if r % 10 == 0:
        print("round %d: %d evaluations, %.0f s" % (r, count, time.time() - t0), flush=True)
for n in (128, 777):
    X, y = synth(n, d=6, seed=n)
    g = gp.Covsum(n, 6)
    g.set_loghyperparam([0.5, 0.5, 0.5])
    g.cg_solve(X, y, budget=60)
    g.close()
b = gp.BCM.split(*synth(24000, d=10, seed=3), 16)
for r in range(rounds):
    hp = np.array([np.log(3.0), 0.0, np.log(0.1)]) + 0.1 * rng.standard_normal(3)
    b.set_BCM_log_hyperparam(hp)
    p = b.loglik_grad()
    b.set_BCM_log_hyperparam(hp + 0.3)
    b.loglik_grad()
    b.set_BCM_log_hyperparam(hp)
    q = b.loglik_grad()
    assert p[0] == q[0] and tuple(p[1]) == tuple(q[1]), (hp, p, q)
b.close()
for g in hs.values():
    g.close()
print("soak ok: %d single evaluations + %d BCM evaluations in %.0f s" % (count, 3 * rounds, time.time() - t0))
