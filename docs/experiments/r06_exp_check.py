"""Which exponential does k_build run?  K[i][0] = exp(-x_i^2 / 2) (one feature, unit hyper-parameters) against an exact
emulation of exp_neg (kernels.hip) in rational arithmetic, under tuning key 12 = 1 and 0.
   python tools/exp_check.py"""
import os, sys
from fractions import Fraction as F
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cugp_amd.gp as gp
from cugp_amd import capi


def fma(a, b, c):
    return float(F(a) * F(b) + F(c))


def exp_neg(x):
    xc = max(x, -800.0)
    kf = float(np.rint(xc * 1.44269504088896338700e+00))
    r = fma(kf, -6.93147180369123816490e-01, xc)
    r = fma(kf, -1.90821492927058770002e-10, r)
    cs = [1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0,
          1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5]
    q = cs[0]
    for c in cs[1:]:
        q = fma(q, r, c)
    e = fma(q * r, r, r)
    return float(np.ldexp(1.0 + e, int(kf)))


n = 400
rng = np.random.default_rng(3)
x = np.concatenate([[0.0], rng.uniform(0.0, 8.0, n - 1)])
arg = -(x * x) * 0.5
emu = np.array([exp_neg(a) for a in arg])
lib = np.exp(arg)
print("emulation differs from libm in %d of %d arguments" % (int(np.sum(emu != lib)), n))
for v in (1, 0):
    capi.check(capi.lib().cugp_set_tuning(12, v))
    g = gp.Covsum(n, 1)
    g.set_data(x.reshape(-1, 1), np.zeros(n))
    g.set_loghyperparam([0.0, 0.0, -50.0])
    K = g.compute_K_train()
    g.close()
    col = K[:, 0].copy()
    col[0] = 1.0
    print("key 12 = %d: GPU column equals the emulation in %d of %d, equals libm in %d" % (v, int(np.sum(col == emu)), n, int(np.sum(col == lib))))
capi.check(capi.lib().cugp_set_tuning(12, 1))
