"""BCM on one GPU with / without the inverse blocks beside the factorisation (tuning key 3), interleaved."""
import os, sys, time, statistics
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cugp_amd.gp as gp
from cugp_amd import capi
from conftest import synth
pipes = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,8,4".split(","))]
for K, rows in ((16, 1500), (4, 6000), (1, 1500), (1, 3000), (1, 4096), (1, 6000)):
    X, y = synth(K * rows, seed=5)
    b = gp.BCM.split(X, y, K)
    hp = np.array([np.log(3.0), 0.0, np.log(0.1)])
    res = {p: [] for p in pipes}
    for it in range(8):
        for p in pipes:
            capi.check(capi.lib().cugp_set_tuning(3, p))
            b.set_BCM_log_hyperparam(hp + 1e-3 * (it * 3 + p))
            t0 = time.perf_counter(); b.loglik_grad(); t1 = time.perf_counter()
            if it > 0:
                res[p].append((t1 - t0) * 1e3)
    print("K=%2d x %5d rows: " % (K, rows) + "  ".join("pipe=%d %.3f ms" % (p, statistics.median(v)) for p, v in res.items()), flush=True)
    b.close()
capi.check(capi.lib().cugp_set_tuning(3, -1))
