"""Interleaved A/B of tuning keys (kernels.h TUNE_*, by number) on one BCM evaluation (LL + gradient) on one GPU:
    python tools/bcm_ab2.py <experts> <rows per expert> base 14=0 2=256,14=1 ..."""
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cugp_amd.gp as gp                                  # noqa: E402
from cugp_amd import capi                                 # noqa: E402
from conftest import synth                                # noqa: E402
DEFAULT = {0: 768, 1: 1200, 2: 384, 3: -1, 4: 511, 5: 1, 6: 1, 7: 1 << 20, 8: 16, 9: 500, 10: 32, 11: 1, 12: 1, 13: 256, 14: 1536}     # as tools/ab2.py

K, rows = int(sys.argv[1]), int(sys.argv[2])
names = sys.argv[3:] or ["base"]
variants = [{} if a == "base" else dict((int(k), int(v)) for k, v in (kv.split("=") for kv in a.split(","))) for a in names]
rounds = int(os.environ.get("AB_ROUNDS", "12"))


def apply(var):
    for k, d in DEFAULT.items():
        capi.check(capi.lib().cugp_set_tuning(k, var.get(k, d)))


X, y = synth(K * rows)
hp = np.array([np.log(3.0), 0.0, np.log(0.1)])
b = gp.BCM.split(X, y, K)
res = [[] for _ in variants]
vals = [None] * len(variants)
for rnd in range(rounds + 2):
    for i, var in enumerate(variants):
        apply(var)
        b.set_BCM_log_hyperparam(hp + (1e-4 * rnd + 1e-6 * i if rnd > 0 else 0.0))
        t0 = time.perf_counter()
        ll, g, per = b.loglik_grad()
        t1 = time.perf_counter()
        if rnd == 0:
            vals[i] = (ll, g)
        elif rnd > 1:
            res[i].append((t1 - t0) * 1e3)
apply({})
for a, r, v in zip(names, res, vals):
    print("K=%d x %d rows  %-18s %.3f ms (min %.3f)   ll %.10f g %s" % (K, rows, a, statistics.median(r), min(r), v[0],
          np.array2string(np.asarray(v[1]), precision=10)), flush=True)
