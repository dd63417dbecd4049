"""Stream priorities (TUNE_STREAM_PRIO, read at handle creation) for GROUPED experts and small single matrices:
a fresh BCM / handle per variant, variants interleaved.   python tools/prio_bcm_ab.py"""
import os, sys, time, statistics
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cugp_amd.gp as gp
from cugp_amd import capi
from conftest import synth
hp = np.array([np.log(3.0), 0.0, np.log(0.1)])
for K, rows in ((16, 1500), (2, 1500), (4, 6000), (2, 8192), (8, 3000), (1, 1500), (1, 4096)):
    X, y = synth(K * rows, seed=5)
    res = {0: [], 1: []}
    for rnd in range(3):
        for mode in (0, 1):
            capi.check(capi.lib().cugp_set_tuning(15, mode))
            b = gp.BCM.split(X, y, K)
            for it in range(3):
                b.set_BCM_log_hyperparam(hp + 1e-3 * it); b.loglik_grad()
            ts = []
            for it in range(8):
                b.set_BCM_log_hyperparam(hp + 1e-4 * it)
                t0 = time.perf_counter(); b.loglik_grad(); ts.append((time.perf_counter() - t0) * 1e3)
            res[mode].append(statistics.median(ts))
            b.close()
    print("K=%2d x %5d rows: default %.3f ms   factorisation stream high %.3f ms" % (K, rows, statistics.median(res[0]), statistics.median(res[1])), flush=True)
capi.check(capi.lib().cugp_set_tuning(15, 0))
