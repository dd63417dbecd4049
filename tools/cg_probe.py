"""Wall time of the host CG loop (cugp_cg_solve) against evaluations x device time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cugp_amd.gp as gp
from conftest import synth
for n, budget in ((8192, 20), (1500, 100)):
    X, y = synth(n)
    g = gp.Covsum(n, 10)
    g.set_data(X, y)
    g.set_loghyperparam(np.array([0.5, 0.5, 0.5]))
    g.loglik_grad()
    t0 = time.perf_counter(); tr = g.cg_solve(None, None, budget); t1 = time.perf_counter()
    print("n=%d: cg_solve %d evaluations in %.1f ms = %.3f ms each; end hp %s f=%.6f" % (
        n, len(tr), (t1 - t0) * 1e3, (t1 - t0) * 1e3 / len(tr), g.get_loghyperparam(), tr[-1, 3]), flush=True)
    g.close()
