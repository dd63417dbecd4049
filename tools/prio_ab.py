"""Stream priorities (TUNE_STREAM_PRIO, read at handle creation) A/B at one size: a fresh handle per variant, the
variants interleaved; wall time per LL+grad evaluation on the default path, then the average duration of the step
launches and their trailing-update rate from a profiled pass (cugp_set_profiling 2).
    python tools/prio_ab.py [n=8192] [rounds=3]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
X, y = synth(n)
hp = np.array([np.log(3.0), 0.0, np.log(0.1)])
res = {m: dict(wall=[], step_us=[], wide_us=[], tu=[]) for m in (0, 1, 2, 3)}
for rnd in range(rounds):
    for mode in (0, 1, 2, 3):
        capi.check(capi.lib().cugp_set_tuning(15, mode))
        g = gp.Covsum(n, 10)
        g.set_data(X, y)
        for it in range(3):
            g.set_loghyperparam(hp + 1e-3 * it)
            g.loglik_grad()
        t0 = time.perf_counter()
        for it in range(10):
            g.set_loghyperparam(hp + 1e-4 * it)
            ll, gr = g.loglik_grad()
        res[mode]["wall"].append((time.perf_counter() - t0) * 100.0)
        g.set_profiling(2)
        g.loglik_grad()
        for kd in range(8):
            g.kernel_stats(reset=True, kind=kd)
        for it in range(16):
            g.set_loghyperparam(hp + 1e-4 * it)
            g.loglik_grad()
        ks, kw = g.kernel_stats(kind=0), g.kernel_stats(kind=1)
        res[mode]["step_us"].append(1e3 * ks["sum_ms"] / max(1, ks["launches"]))
        res[mode]["wide_us"].append(1e3 * kw["sum_ms"] / max(1, kw["launches"]))
        res[mode]["tu"].append((kw["flop"] + 16 * ks["flop"]) / ((kw["sum_ms"] + 16 * ks["sum_ms"]) * 1e-3) / 1e12)
        g.close()
capi.check(capi.lib().cugp_set_tuning(15, 0))
for mode, name in ((0, "default priorities"), (1, "factorisation high"), (2, "inverse streams low"), (3, "both")):
    r = res[mode]
    print("%-22s eval %.3f ms (min %.3f)   k_syrk_step %.1f us avg   k_syrk_wide %.0f us   trailing update %.1f TF/s (%.3f of peak)   ll %.10f"
          % (name, statistics.median(r["wall"]), min(r["wall"]), statistics.median(r["step_us"]), statistics.median(r["wide_us"]),
             statistics.median(r["tu"]), statistics.median(r["tu"]) / 78.6, ll), flush=True)
