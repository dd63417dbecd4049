"""BCM on one GPU: K experts evaluated concurrently (own streams) vs one after another."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cugp_amd.gp as gp
from conftest import synth
for K, rows in ((16, 1500), (4, 6000), (2, 1500), (2, 8192)):
    X, y = synth(K * rows, seed=5)
    b = gp.BCM.split(X, y, K)
    hp = np.array([np.log(3.0), 0.0, np.log(0.1)])
    for it in range(3):
        b.set_BCM_log_hyperparam(hp + 1e-3 * it)
        t0 = time.perf_counter(); ll, g, per = b.loglik_grad(); t1 = time.perf_counter()
    one = gp.Covsum(rows, 10); one.set_data(X[:rows], y[:rows])
    for it in range(3):
        one.set_loghyperparam(hp + 1e-3 * it)
        s0 = time.perf_counter(); one.loglik_grad(); s1 = time.perf_counter()
    print("K=%2d x %5d rows: concurrent %.2f ms  (one expert alone %.2f ms -> serial %.2f ms)  %.1f expert-evals/s" % (
        K, rows, (t1 - t0) * 1e3, (s1 - s0) * 1e3, K * (s1 - s0) * 1e3, K / (t1 - t0)), flush=True)
    b.close(); one.close()
