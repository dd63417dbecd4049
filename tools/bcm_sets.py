"""One BCM evaluation (LL + gradient) on ONE GPU with the experts in 1, 2, 4 ... independent sets (a device listed
several times in cugp_bcm_create_multi gives every listing its own group of shared launches and its own streams):
    python tools/bcm_sets.py <experts> <rows per expert> [sets ...]"""
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cugp_amd.gp as gp                                  # noqa: E402
from conftest import synth                                # noqa: E402

K, rows = int(sys.argv[1]), int(sys.argv[2])
sets = [int(a) for a in sys.argv[3:]] or [1, 2, 4]
X, y = synth(K * rows)
hp = np.array([np.log(3.0), 0.0, np.log(0.1)])
bs = [gp.BCM.split(X, y, K, devices=[0] * s) for s in sets]
res = [[] for _ in sets]
vals = [None] * len(sets)
for rnd in range(12):
    for i, b in enumerate(bs):
        b.set_BCM_log_hyperparam(hp + 1e-4 * rnd)
        t0 = time.perf_counter()
        ll, g, per = b.loglik_grad()
        t1 = time.perf_counter()
        if rnd == 0:
            vals[i] = (ll, g)
        elif rnd > 1:
            res[i].append((t1 - t0) * 1e3)
for s, r, v in zip(sets, res, vals):
    print("K=%d x %d rows, %d set(s): %.3f ms (min %.3f)   ll %.10f g %s" % (K, rows, s, statistics.median(r), min(r), v[0],
          np.array2string(np.asarray(v[1]), precision=10)), flush=True)
