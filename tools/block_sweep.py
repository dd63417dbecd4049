"""Sweep of hand-over block sizes (TUNE_PIPE_BLOCK) over matrix sizes: the overlapped evaluation (k_trtri_block for every
block of w tiles and the last, larger or smaller one) against the single-stream evaluation of the same handle: LL and
gradient agree to rounding (the partition changes summation orders), K^-1 K = I on the factor's own matrix.
    python tools/block_sweep.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cugp_amd.gp as gp                                  # noqa: E402
from cugp_amd import capi                                 # noqa: E402
from conftest import synth                                # noqa: E402

hp = np.array([np.log(3.0), 0.0, np.log(0.1)])
bad = 0
for n in (130, 300, 515, 900, 1100, 1500, 1700, 2100, 2500):
    X, y = synth(n, d=7, seed=n)
    g = gp.Covsum(n, 7)
    g.set_data(X, y)
    g.set_loghyperparam(hp)
    g.set_overlap(False)
    ll0, g0 = g.loglik_grad()
    K0 = g.get_K_inverse()
    g.set_overlap(True)
    for w in (1, 2, 3, 4, 5, 6, 7, 8, 11, 16):
        capi.check(capi.lib().cugp_set_tuning(3, w))
        g.set_loghyperparam(hp + 1e-9)
        g.loglik_grad()
        g.set_loghyperparam(hp)
        ll, gr = g.loglik_grad()
        Ki = g.get_K_inverse()
        e_ll = abs(ll - ll0) / max(1.0, abs(ll0))
        e_g = np.max(np.abs(gr - g0) / (np.abs(g0) + 1e-9 * np.max(np.abs(g0))))
        e_k = np.max(np.abs(Ki - K0)) / np.max(np.abs(K0))
        ok = e_ll < 1e-12 and e_g < 1e-9 and e_k < 1e-11 and np.isfinite(ll)
        bad += 0 if ok else 1
        if not ok or w in (1, 16):
            print("n %5d (%2d tiles) w %2d: dLL %.1e dgrad %.1e dKinv %.1e %s" % (n, (n + 127) // 128, w, e_ll, e_g, e_k, "" if ok else "<-- BAD"), flush=True)
    g.close()
capi.check(capi.lib().cugp_set_tuning(3, -1))
print("block sweep: %s" % ("ok" if bad == 0 else "%d mismatches" % bad))
