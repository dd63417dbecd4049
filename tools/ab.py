"""Interleaved A/B of launch-shape knobs in ONE process on ONE device (timings across gpurun calls land on
different boards and are not comparable).   python tools/ab.py <n> key=val,key=val ... (each arg one variant)"""
import os
import sys
import statistics

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cugp_amd.gp as gp                                  # noqa: E402
from cugp_amd import capi                                 # noqa: E402
from conftest import synth                                # noqa: E402

KEYS = {"lauum": 0, "trtri": 1, "syrk": 2, "pipe": 3, "bwm2": 4}
DEFAULT = {0: 768, 1: 1200, 2: 384, 3: -1, 4: 511}
n = int(sys.argv[1])
variants = [dict((KEYS[k], int(v)) for k, v in (kv.split("=") for kv in a.split(","))) if a != "base" else {}
            for a in sys.argv[2:]]
X, y = synth(n)
g = gp.Covsum(n, 10)
g.set_data(X, y)
g.set_profiling(1)
hp = np.array([np.log(3.0), 0.0, np.log(0.1)])
res = [dict(potrf=[], trtri=[], lauum=[], total=[]) for _ in variants]
for rnd in range(7):
    for vi, var in enumerate(variants):
        for k, d in DEFAULT.items():
            capi.lib().cugp_set_tuning(k, var.get(k, d))
        g.set_loghyperparam(hp + 1e-4 * (rnd * len(variants) + vi))
        g.loglik_grad()
        ph = g.phase_ms()
        if rnd > 0:
            for k in res[vi]:
                res[vi][k].append(ph[k])
for a, r in zip(sys.argv[2:], res):
    print("%-28s " % a + "  ".join("%s %.3f (min %.3f)" % (k, statistics.median(v), min(v)) for k, v in r.items()), flush=True)
