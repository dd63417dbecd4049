import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import cugp_amd.gp as gp
from conftest import synth
hp = np.array([np.log(3.0), 0.0, np.log(0.1)])
for n in (1500, 4096, 8192):
    X, y = synth(n)
    g = gp.Covsum(n, 10); g.set_data(X, y)
    for mode in ("LL only", "LL+grad"):
        ts = []
        for it in range(12):
            g.set_loghyperparam(hp + 1e-4 * it)
            t0 = time.perf_counter()
            if mode == "LL only": g.compute_loglikelihood()
            else: g.loglik_grad()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("n=%5d %-8s median %.3f ms  min %.3f" % (n, mode, sorted(ts[2:])[len(ts[2:]) // 2], min(ts[2:])), flush=True)
    g.set_profiling(1)
    g.set_loghyperparam(hp + 0.01); g.compute_loglikelihood(); print("   phases LL-only:", g.phase_ms())
    g.close()
