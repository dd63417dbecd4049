"""Interleaved A/B of tuning keys (kernels.h TUNE_*, by number) in ONE process on ONE device: whole LL+grad
evaluation (wall clock and phases), the factorisation alone (cugp_bench_la op 0) and the three LA phases together
(op 3), plus the results of every variant so a rounding-order change is visible next to its timing.
    python tools/ab2.py 8192 base 8=1 8=2 8=4,9=0 12=0"""
import ctypes as C
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

DEFAULT = {0: 768, 1: 1200, 2: 384, 3: -1, 4: 511, 5: 1, 6: 1, 7: 1 << 20, 8: 16, 9: 500, 10: 32, 11: 1, 12: 2100, 13: 256, 14: 1536, 15: 0, 16: 1 << 21, 17: 1, 18: 1}
n = int(sys.argv[1])
names = sys.argv[2:] or ["base"]
variants = [{} if a == "base" else dict((int(k), int(v)) for k, v in (kv.split("=") for kv in a.split(","))) for a in names]
rounds = int(os.environ.get("AB_ROUNDS", "6"))
la = os.environ.get("AB_LA", "1") != "0"


def apply(var):
    for k, d in DEFAULT.items():
        capi.check(capi.lib().cugp_set_tuning(k, var.get(k, d)))


X, y = synth(n)
g = gp.Covsum(n, 10)
g.set_data(X, y)
PROF = int(os.environ.get('AB_PROF', '1'))   # 0: the default path (wall clock only)
g.set_profiling(PROF)
hp = np.array([np.log(3.0), 0.0, np.log(0.1)])
res = [dict(wall=[], potrf=[], total=[], la0=[], la3=[]) for _ in variants]
vals = [None] * len(variants)
seq = os.environ.get("AB_SEQ", "0") != "0"       # variant-major order: for keys that rebuild streams when they change
order = ([(rnd, vi) for vi in range(len(variants)) for rnd in range(-1, rounds + 1)] if seq
         else [(rnd, vi) for rnd in range(rounds + 1) for vi in range(len(variants))])
for rnd, vi in order:
    var = variants[vi]
    apply(var)
    g.set_loghyperparam(hp + (1e-4 * rnd if rnd > 0 else 0.0) + (1e-6 * vi if rnd > 0 else 0.0) - (1e-5 if rnd < 0 else 0.0))
    t0 = time.perf_counter()
    ll, gr = g.loglik_grad()
    t1 = time.perf_counter()
    ph = g.phase_ms() if PROF else {'potrf': float('nan'), 'total': float('nan')}
    if rnd < 0:
        continue                                  # warm-up after a switch (sequential mode)
    if rnd == 0:
        vals[vi] = (ll, gr)
        continue
    res[vi]["wall"].append((t1 - t0) * 1e3)
    res[vi]["potrf"].append(ph["potrf"])
    res[vi]["total"].append(ph["total"])
if la:
    for vi, var in enumerate(variants):
        apply(var)
        for op, key in ((0, "la0"), (3, "la3")):
            ms = C.c_double()
            capi.check(capi.lib().cugp_bench_la(op, n, 0, 4, C.byref(ms)))
            res[vi][key].append(ms.value)
apply({})
for a, r, v in zip(names, res, vals):
    print("%-22s eval wall %.3f (min %.3f) dev %.3f  potrf-phase %.3f | alone: potrf %s  all3 %s | ll %.10f g %s" % (
        a, statistics.median(r["wall"]), min(r["wall"]), statistics.median(r["total"]), statistics.median(r["potrf"]),
        "%.3f" % r["la0"][0] if r["la0"] else "-", "%.3f" % r["la3"][0] if r["la3"] else "-",
        v[0], np.array2string(v[1], precision=10)), flush=True)
