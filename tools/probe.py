"""GPU probe: MFMA peak, tile-product check, small parity, phase timings at a few sizes."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cugp_amd.gp as gp                                  # noqa: E402
from conftest import synth                                # noqa: E402

print("fp64 MFMA peak probe: %.1f TFLOP/s" % gp.mfma_peak_tflops(), flush=True)
rng = np.random.default_rng(0)
A = rng.standard_normal((256, 160)); B = rng.standard_normal((384, 160))
print("gemm max err", np.max(np.abs(gp.test_gemm_nt(A, B) - A @ B.T)), flush=True)

sizes = [int(s) for s in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1024,4096,8192".split(","))]
for n in sizes:
    X, y = synth(n)
    g = gp.Covsum(n, 10)
    g.set_data(X, y)
    g.set_profiling(2)
    hp = np.array([np.log(3.0), 0.0, np.log(0.1)])
    for it in range(3):
        g.set_loghyperparam(hp + 1e-3 * it)
        t0 = time.time()
        ll, gr = g.loglik_grad()
        t1 = time.time()
        ph = g.phase_ms()
    ks = g.kernel_stats(reset=True)
    print("n=%d ll=%.6f grad=%s wall=%.2f ms phases=%s" % (n, ll, gr, (t1 - t0) * 1e3,
          {k: round(v, 3) for k, v in ph.items()}), flush=True)
    if ks["launches"]:
        print("   syrk: %d launches, %.3f ms total, %.2f TFLOP/s" % (ks["launches"], ks["sum_ms"],
              ks["flop"] / ks["sum_ms"] / 1e9), flush=True)
    npad = -(-n // 128) * 128
    print("   potrf %.2f TF/s, trtri %.2f, lauum %.2f, eval %.2f TF/s (N^3 flop)" % (
        npad ** 3 / 3 / ph["potrf"] / 1e9, npad ** 3 / 3 / ph["trtri"] / 1e9, npad ** 3 / 3 / ph["lauum"] / 1e9,
        n ** 3 / ph["total"] / 1e9), flush=True)
    g.set_loghyperparam(hp + 0.5)
    t0 = time.time(); ll = g.compute_loglikelihood(); t1 = time.time()
    print("   LL-only wall %.2f ms" % ((t1 - t0) * 1e3), flush=True)
    g.close()
