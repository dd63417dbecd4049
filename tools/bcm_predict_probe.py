"""BCM prediction timing on one GPU: K experts, nt test points (tools/bcm_predict_probe.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cugp_amd.gp as gp
from conftest import synth
hp = np.array([np.log(3.0), 0.0, np.log(0.1)])
for K, rows in ((16, 1500), (4, 6000), (2, 1500)):
    X, y = synth(K * rows, seed=5)
    Xt = synth(1000, seed=9)[0]
    b = gp.BCM.split(X, y, K)
    b.set_BCM_log_hyperparam(hp)
    b.loglik_grad()
    m0, v0 = b.compute_BCM_test_means_and_var(Xt)
    ts = []
    for it in range(6):
        t0 = time.perf_counter(); m, v = b.compute_BCM_test_means_and_var(Xt); ts.append((time.perf_counter() - t0) * 1e3)
    assert np.array_equal(m, m0) and np.array_equal(v, v0)
    tc = []
    for it in range(4):                       # with new hyper-parameters in front: factor + inverse of every expert first
        b.set_BCM_log_hyperparam(hp + 1e-3 * (it + 1))
        t0 = time.perf_counter(); b.compute_BCM_test_means_and_var(Xt); tc.append((time.perf_counter() - t0) * 1e3)
    print("K=%2d x %5d rows, 1000 test points: predict %.3f ms (min %.3f); after new hyper-parameters %.3f ms; checksum %.12g %.12g"
          % (K, rows, sorted(ts)[len(ts) // 2], min(ts), sorted(tc)[len(tc) // 2], float(np.sum(m0)), float(np.sum(v0))), flush=True)
    b.close()
