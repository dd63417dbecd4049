import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cugp_amd.gp as gp
from conftest import synth
for n, nt in ((8192, 1000), (8192, 128), (1500, 1000)):
    X, y = synth(n)
    Xt = synth(nt, seed=9)[0]
    g = gp.Covsum(n, 10)
    g.set_data(X, y)
    g.set_loghyperparam(np.array([np.log(3.0), 0.0, np.log(0.1)]))
    g.loglik_grad()
    ts = []
    for i in range(5):
        t0 = time.perf_counter(); m, v = g.compute_test_means_and_variances(None, None, Xt); ts.append(time.perf_counter() - t0)
    print("n=%d nt=%d predict %.3f ms (min of 5; includes H2D of Xt and D2H of mean/var)" % (n, nt, min(ts) * 1e3), flush=True)
    g.close()
