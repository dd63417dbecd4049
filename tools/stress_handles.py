import sys, time, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cugp_amd.gp as gp
from conftest import synth
t0 = time.time()
for it in range(150):
    n = [130, 300, 700, 1500][it % 4]
    X, y = synth(n, d=4, seed=it)
    g = gp.Covsum(n, 4)
    g.set_loghyperparam([0.5, 0.1, -1.0])
    ll, gr = g.loglik_grad(X, y)
    assert np.isfinite(ll)
    if it % 10 == 0:
        b = gp.BCM.split(X, y, 3)
        b.set_BCM_log_hyperparam([0.5, 0.1, -1.0]); b.loglik_grad(); b.close()
    g.close()
print("150 create/evaluate/destroy cycles ok in %.1f s" % (time.time() - t0))
