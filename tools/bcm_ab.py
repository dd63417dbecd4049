"""BCM on one GPU: interleaved A/B of tuning keys (kernels.h TUNE_*), e.g.
   python tools/bcm_ab.py 5=0 5=1        # launch by launch vs captured graph
   python tools/bcm_ab.py 3=0 3=-1       # (single experts) inverse after / beside the factorisation"""
import os, sys, time, statistics
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cugp_amd.gp as gp
from cugp_amd import capi
from conftest import synth
DEFAULT = {0: 768, 1: 1200, 2: 384, 3: -1, 4: 511, 5: 1, 6: 1, 7: 1048576}
variants = [dict((int(k), int(v)) for k, v in (kv.split("=") for kv in a.split(","))) for a in sys.argv[1:]] or [{}]
for K, rows in ((16, 1500), (2, 1500), (4, 6000), (2, 8192), (8, 3000), (1, 1500), (1, 256)):
    X, y = synth(K * rows, seed=5)
    b = gp.BCM.split(X, y, K)
    hp = np.array([np.log(3.0), 0.0, np.log(0.1)])
    res = [[] for _ in variants]
    for it in range(8):
        for vi, var in enumerate(variants):
            for k, d in DEFAULT.items():
                capi.check(capi.lib().cugp_set_tuning(k, var.get(k, d)))
            b.set_BCM_log_hyperparam(hp + 1e-3 * (it * len(variants) + vi))
            t0 = time.perf_counter(); b.loglik_grad(); t1 = time.perf_counter()
            if it > 1:
                res[vi].append((t1 - t0) * 1e3)
    print("K=%2d x %5d rows: " % (K, rows) + "  ".join("%s %.3f ms" % (a, statistics.median(v))
          for a, v in zip(sys.argv[1:] or ["default"], res)), flush=True)
    b.close()
for k, d in DEFAULT.items():
    capi.check(capi.lib().cugp_set_tuning(k, d))
