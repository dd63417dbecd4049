"""ctypes bindings for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  It wraps

* ``oracle/libgp_oracle.so``  -- our plain-C restatement (``gp_oracle.c``), and
* ``oracle/_ref/libref_*.so`` -- the reference's own sources compiled where they
  lie (``oracle/Makefile``; present only where that build was possible).

Nothing in ``cugp_amd`` imports from here.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def build(ref=True):
    """(Re)build the restatement and, where the reference checkout exists, oracle/_ref."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


_OBJ = C.CFUNCTYPE(None, C.c_void_p, _dp, _dp, _dp)


class Oracle:
    """Restatement of Covsum / BCM / matrixops (gp_oracle.h)."""

    def __init__(self):
        path = os.path.join(HERE, "libgp_oracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = self.lib = C.CDLL(path)
        L.oracle_gp_create.restype = C.c_void_p
        L.oracle_gp_create.argtypes = [C.c_int, C.c_int]
        L.oracle_gp_destroy.argtypes = [C.c_void_p]
        L.oracle_gp_set_loghyper.argtypes = [C.c_void_p, _dp]
        L.oracle_gp_get_loghyper.argtypes = [C.c_void_p, _dp]
        L.oracle_gp_K_train.argtypes = [C.c_void_p, _dp, _dp]
        L.oracle_gp_k_test.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.oracle_gp_sqdist.argtypes = [C.c_void_p, _dp, C.c_double, _dp]
        L.oracle_gp_loglik.restype = C.c_double
        L.oracle_gp_loglik.argtypes = [C.c_void_p, _dp, _dp]
        L.oracle_gp_grad.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.oracle_gp_predict.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_int, _dp, _dp]
        L.oracle_nlpp.restype = C.c_double
        L.oracle_nlpp.argtypes = [_dp, _dp, _dp, C.c_int]
        L.oracle_gp_cg_solve.restype = C.c_int
        L.oracle_gp_cg_solve.argtypes = [C.c_void_p, _dp, _dp, C.c_int, _dp, C.c_int]
        L.oracle_gp_rprop_solve.restype = C.c_int
        L.oracle_gp_rprop_solve.argtypes = [C.c_void_p, _dp, _dp, C.c_int, _dp, C.c_int]
        L.oracle_cg_minimize.restype = C.c_int
        L.oracle_cg_minimize.argtypes = [_OBJ, C.c_void_p, _dp, C.c_int, _dp, C.c_int]
        L.oracle_rprop_minimize.restype = C.c_int
        L.oracle_rprop_minimize.argtypes = [_OBJ, C.c_void_p, _dp, C.c_int, _dp, C.c_int]
        L.oracle_get_cholesky.argtypes = [_dp, _dp, C.c_int]
        L.oracle_chol_and_det.argtypes = [_dp, _dp, C.c_int, _dp, _dp]
        L.oracle_K_inverse.argtypes = [_dp, _dp, C.c_int]
        L.oracle_Kinvy.argtypes = [_dp, _dp, _dp, C.c_int]
        L.oracle_bcm_create.restype = C.c_void_p
        L.oracle_bcm_create.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_int]
        L.oracle_bcm_destroy.argtypes = [C.c_void_p]
        L.oracle_bcm_expert_rows.restype = C.c_int
        L.oracle_bcm_expert_rows.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.oracle_bcm_set_loghyper.argtypes = [C.c_void_p, _dp]
        L.oracle_bcm_get_loghyper.argtypes = [C.c_void_p, _dp]
        L.oracle_bcm_loglik.restype = C.c_double
        L.oracle_bcm_loglik.argtypes = [C.c_void_p, _dp]
        L.oracle_bcm_grad.argtypes = [C.c_void_p, _dp]
        L.oracle_bcm_predict.argtypes = [C.c_void_p, _dp, C.c_int, _dp, _dp]
        L.oracle_poe.argtypes = [_dp, _dp, C.c_int, C.c_int, _dp, _dp]
        L.oracle_bcm_cg_solve.restype = C.c_int
        L.oracle_bcm_cg_solve.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int]

    # ---- single GP ----
    def _gp(self, X, hp):
        X = _f64(X)
        g = self.lib.oracle_gp_create(X.shape[0], X.shape[1])
        self.lib.oracle_gp_set_loghyper(g, _p(_f64(hp)))
        return g, X

    def K_train(self, X, hp):
        g, X = self._gp(X, hp)
        K = np.empty((X.shape[0], X.shape[0]))
        self.lib.oracle_gp_K_train(g, _p(X), _p(K))
        self.lib.oracle_gp_destroy(g)
        return K

    def k_test(self, X, hp, xt):
        g, X = self._gp(X, hp)
        out = np.empty(X.shape[0])
        self.lib.oracle_gp_k_test(g, _p(X), _p(_f64(xt)), _p(out))
        self.lib.oracle_gp_destroy(g)
        return out

    def sqdist(self, X, c):
        g, X = self._gp(X, [0, 0, 0])
        S = np.empty((X.shape[0], X.shape[0]))
        self.lib.oracle_gp_sqdist(g, _p(X), float(c), _p(S))
        self.lib.oracle_gp_destroy(g)
        return S

    def loglik(self, X, y, hp):
        g, X = self._gp(X, hp)
        ll = self.lib.oracle_gp_loglik(g, _p(X), _p(_f64(y)))
        self.lib.oracle_gp_destroy(g)
        return ll

    def grad(self, X, y, hp):
        g, X = self._gp(X, hp)
        out = np.empty(3)
        self.lib.oracle_gp_grad(g, _p(X), _p(_f64(y)), _p(out))
        self.lib.oracle_gp_destroy(g)
        return out

    def loglik_grad(self, X, y, hp):
        return self.loglik(X, y, hp), self.grad(X, y, hp)

    def predict(self, X, y, hp, Xt):
        g, X = self._gp(X, hp)
        Xt = _f64(Xt)
        m, v = np.empty(Xt.shape[0]), np.empty(Xt.shape[0])
        self.lib.oracle_gp_predict(g, _p(X), _p(_f64(y)), _p(Xt), Xt.shape[0], _p(m), _p(v))
        self.lib.oracle_gp_destroy(g)
        return m, v

    def nlpp(self, actual, mean, var):
        a, m, v = _f64(actual), _f64(mean), _f64(var)
        return self.lib.oracle_nlpp(_p(a), _p(m), _p(v), a.shape[0])

    def cg_solve(self, X, y, hp, budget=100):
        """Returns (final_hp, trace[n_evals,4]) -- trace rows are [hp0,hp1,hp2,f=-LL]."""
        g, X = self._gp(X, hp)
        tr = np.zeros((4 * budget + 8, 4))
        ne = self.lib.oracle_gp_cg_solve(g, _p(X), _p(_f64(y)), budget, _p(tr), tr.shape[0])
        out = np.empty(3)
        self.lib.oracle_gp_get_loghyper(g, _p(out))
        self.lib.oracle_gp_destroy(g)
        return out, tr[:ne]

    def rprop_solve(self, X, y, hp, iters=100):
        g, X = self._gp(X, hp)
        tr = np.zeros((2 * iters + 8, 4))
        ne = self.lib.oracle_gp_rprop_solve(g, _p(X), _p(_f64(y)), iters, _p(tr), tr.shape[0])
        out = np.empty(3)
        self.lib.oracle_gp_get_loghyper(g, _p(out))
        self.lib.oracle_gp_destroy(g)
        return out, tr[:ne]

    def cg_minimize(self, fn, theta, budget=100):
        """fn(theta ndarray[3]) -> (f, g[3]); generic objective (host-logic tests)."""
        def cb(_ctx, th, f, g):
            fv, gv = fn(np.array([th[0], th[1], th[2]]))
            f[0] = fv
            for i in range(3):
                g[i] = gv[i]
        th = _f64(theta).copy()
        tr = np.zeros((4 * budget + 8, 4))
        ne = self.lib.oracle_cg_minimize(_OBJ(cb), None, _p(th), budget, _p(tr), tr.shape[0])
        return th, tr[:ne]

    def rprop_minimize(self, fn, theta, iters=100):
        def cb(_ctx, th, f, g):
            fv, gv = fn(np.array([th[0], th[1], th[2]]))
            f[0] = fv
            for i in range(3):
                g[i] = gv[i]
        th = _f64(theta).copy()
        tr = np.zeros((2 * iters + 8, 4))
        ne = self.lib.oracle_rprop_minimize(_OBJ(cb), None, _p(th), iters, _p(tr), tr.shape[0])
        return th, tr[:ne]

    # ---- LA ----
    def cholesky(self, K):
        K = _f64(K)
        L = np.empty_like(K)
        self.lib.oracle_get_cholesky(_p(K), _p(L), K.shape[0])
        return L

    def chol_and_det(self, K, y):
        K = _f64(K)
        q, d = C.c_double(), C.c_double()
        self.lib.oracle_chol_and_det(_p(K), _p(_f64(y)), K.shape[0], C.byref(q), C.byref(d))
        return q.value, d.value

    def K_inverse(self, K):
        K = _f64(K)
        o = np.empty_like(K)
        self.lib.oracle_K_inverse(_p(K), _p(o), K.shape[0])
        return o

    def Kinvy(self, K, y):
        K = _f64(K)
        o = np.empty(K.shape[0])
        self.lib.oracle_Kinvy(_p(K), _p(_f64(y)), _p(o), K.shape[0])
        return o

    # ---- BCM ----
    def bcm(self, X, y, K, hp):
        return OracleBCM(self, X, y, K, hp)

    def poe(self, means, vars_):
        means, vars_ = _f64(means), _f64(vars_)
        K, nt = means.shape
        m, v = np.empty(nt), np.empty(nt)
        self.lib.oracle_poe(_p(means), _p(vars_), K, nt, _p(m), _p(v))
        return m, v


class OracleBCM:
    def __init__(self, o, X, y, K, hp):
        self.o, self.X, self.y = o, _f64(X), _f64(y)
        self.K = K
        self.h = o.lib.oracle_bcm_create(_p(self.X), _p(self.y), self.X.shape[0], self.X.shape[1], K)
        self.set_loghyper(hp)

    def set_loghyper(self, hp):
        self.hp = _f64(hp)
        self.o.lib.oracle_bcm_set_loghyper(self.h, _p(self.hp))

    def expert_rows(self, k):
        off = C.c_int()
        n = self.o.lib.oracle_bcm_expert_rows(self.h, k, C.byref(off))
        return off.value, n

    def loglik(self):
        per = np.empty(self.K)
        return self.o.lib.oracle_bcm_loglik(self.h, _p(per)), per

    def grad(self):
        g = np.empty(3)
        self.o.lib.oracle_bcm_grad(self.h, _p(g))
        return g

    def predict(self, Xt):
        Xt = _f64(Xt)
        m, v = np.empty(Xt.shape[0]), np.empty(Xt.shape[0])
        self.o.lib.oracle_bcm_predict(self.h, _p(Xt), Xt.shape[0], _p(m), _p(v))
        return m, v

    def cg_solve(self, budget=100):
        tr = np.zeros((4 * budget + 8, 4))
        ne = self.o.lib.oracle_bcm_cg_solve(self.h, budget, _p(tr), tr.shape[0])
        final = np.empty(3)
        self.o.lib.oracle_bcm_get_loghyper(self.h, _p(final))     # the kept (best) point, not the last probe
        return final, tr[:ne]

    def close(self):
        if self.h:
            self.o.lib.oracle_bcm_destroy(self.h)
            self.h = None


class Reference:
    """The reference's own code (oracle/_ref); raises FileNotFoundError when not built."""

    def __init__(self):
        s = os.path.join(HERE, "_ref", "libref_serial.so")
        b = os.path.join(HERE, "_ref", "libref_bcm.so")
        if not (os.path.exists(s) and os.path.exists(b)):
            raise FileNotFoundError("oracle/_ref not built (needs the reference checkout; `make -C oracle ref`)")
        S = self.s = C.CDLL(s)
        B = self.b = C.CDLL(b)
        S.ref_gp_create.restype = C.c_void_p
        S.ref_gp_create.argtypes = [C.c_int, C.c_int]
        S.ref_gp_set_loghyper.argtypes = [C.c_void_p, _dp]
        S.ref_gp_get_loghyper.argtypes = [C.c_void_p, _dp]
        S.ref_gp_loglik.restype = C.c_double
        S.ref_gp_loglik.argtypes = [C.c_void_p, _dp, _dp]
        S.ref_gp_grad.argtypes = [C.c_void_p, _dp, _dp, _dp]
        S.ref_gp_K_train.argtypes = [C.c_void_p, _dp, _dp]
        S.ref_gp_k_test.argtypes = [C.c_void_p, _dp, _dp, _dp]
        S.ref_gp_cg_solve.argtypes = [C.c_void_p, _dp, _dp, C.c_char_p]
        S.ref_gp_rprop_solve.argtypes = [C.c_void_p, _dp, _dp, C.c_char_p]
        S.ref_gp_nlpp.restype = C.c_double
        S.ref_gp_nlpp.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_int]
        S.ref_get_cholesky.argtypes = [_dp, _dp, C.c_int]
        S.ref_K_inverse.argtypes = [_dp, _dp, C.c_int]
        S.ref_chol_and_det.argtypes = [_dp, _dp, C.c_int, _dp, _dp]
        B.ref_bcm_create.restype = C.c_void_p
        B.ref_bcm_create.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_int]
        B.ref_bcm_set_loghyper.argtypes = [C.c_void_p, _dp]
        B.ref_bcm_get_loghyper.argtypes = [C.c_void_p, _dp]
        B.ref_bcm_loglik.restype = C.c_double
        B.ref_bcm_loglik.argtypes = [C.c_void_p, C.c_char_p]
        B.ref_bcm_grad.argtypes = [C.c_void_p, _dp]
        B.ref_bcm_predict.argtypes = [C.c_void_p, _dp, C.c_int, _dp, _dp]
        B.ref_bcm_nlpp.restype = C.c_double
        B.ref_bcm_nlpp.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_int]
        B.ref_bcm_cg_solve.argtypes = [C.c_void_p, C.c_char_p]
        B.ref_gp_predict.argtypes = [_dp, _dp, C.c_int, C.c_int, _dp, _dp, C.c_int, _dp, _dp]

    def _gp(self, X, hp):
        X = _f64(X)
        g = self.s.ref_gp_create(X.shape[0], X.shape[1])
        self.s.ref_gp_set_loghyper(g, _p(_f64(hp)))
        return g, X   # leaked on purpose: the reference's destructor uses delete on new[] memory

    def loglik(self, X, y, hp):
        g, X = self._gp(X, hp)
        return self.s.ref_gp_loglik(g, _p(X), _p(_f64(y)))

    def grad(self, X, y, hp):
        g, X = self._gp(X, hp)
        out = np.empty(3)
        self.s.ref_gp_grad(g, _p(X), _p(_f64(y)), _p(out))
        return out

    def K_train(self, X, hp):
        g, X = self._gp(X, hp)
        K = np.zeros((X.shape[0], X.shape[0]))
        self.s.ref_gp_K_train(g, _p(X), _p(K))
        return K

    def k_test(self, X, hp, xt):
        g, X = self._gp(X, hp)
        out = np.empty(X.shape[0])
        self.s.ref_gp_k_test(g, _p(X), _p(_f64(xt)), _p(out))
        return out

    def cg_solve(self, X, y, hp, logpath):
        g, X = self._gp(X, hp)
        self.s.ref_gp_cg_solve(g, _p(X), _p(_f64(y)), logpath.encode())
        out = np.empty(3)
        self.s.ref_gp_get_loghyper(g, _p(out))
        return out

    def rprop_solve(self, X, y, hp, logpath):
        g, X = self._gp(X, hp)
        self.s.ref_gp_rprop_solve(g, _p(X), _p(_f64(y)), logpath.encode())
        out = np.empty(3)
        self.s.ref_gp_get_loghyper(g, _p(out))
        return out

    def nlpp(self, actual, mean, var):
        g = self.s.ref_gp_create(1, 1)
        a, m, v = _f64(actual), _f64(mean), _f64(var)
        return self.s.ref_gp_nlpp(g, _p(a), _p(m), _p(v), a.shape[0])

    def cholesky(self, K):
        K = _f64(K)
        L = np.empty_like(K)
        self.s.ref_get_cholesky(_p(K), _p(L), K.shape[0])
        return L

    def K_inverse(self, K):
        K = _f64(K)
        o = np.empty_like(K)
        self.s.ref_K_inverse(_p(K), _p(o), K.shape[0])
        return o

    def chol_and_det(self, K, y):
        K = _f64(K)
        q, d = C.c_double(), C.c_double()
        self.s.ref_chol_and_det(_p(K), _p(_f64(y)), K.shape[0], C.byref(q), C.byref(d))
        return q.value, d.value

    def predict(self, X, y, hp, Xt):
        X, Xt = _f64(X), _f64(Xt)
        m, v = np.empty(Xt.shape[0]), np.empty(Xt.shape[0])
        self.b.ref_gp_predict(_p(X), _p(_f64(y)), X.shape[0], X.shape[1], _p(_f64(hp)),
                              _p(Xt), Xt.shape[0], _p(m), _p(v))
        return m, v

    def bcm(self, X, y, K, hp):
        return RefBCM(self, X, y, K, hp)


class RefBCM:
    def __init__(self, r, X, y, K, hp):
        self.r, self.X, self.y, self.K = r, _f64(X), _f64(y), K
        self.h = r.b.ref_bcm_create(_p(self.X), _p(self.y), self.X.shape[0], self.X.shape[1], K)
        self.set_loghyper(hp)

    def set_loghyper(self, hp):
        self.r.b.ref_bcm_set_loghyper(self.h, _p(_f64(hp)))

    def get_loghyper(self):
        out = np.empty(3)
        self.r.b.ref_bcm_get_loghyper(self.h, _p(out))
        return out

    def loglik(self, logpath=None):
        return self.r.b.ref_bcm_loglik(self.h, logpath.encode() if logpath else None)

    def grad(self):
        g = np.empty(3)
        self.r.b.ref_bcm_grad(self.h, _p(g))
        return g

    def predict(self, Xt):
        Xt = _f64(Xt)
        m, v = np.empty(Xt.shape[0]), np.empty(Xt.shape[0])
        self.r.b.ref_bcm_predict(self.h, _p(Xt), Xt.shape[0], _p(m), _p(v))
        return m, v

    def nlpp(self, actual, mean, var):
        a, m, v = _f64(actual), _f64(mean), _f64(var)
        return self.r.b.ref_bcm_nlpp(self.h, _p(a), _p(m), _p(v), a.shape[0])

    def cg_solve(self, logpath):
        self.r.b.ref_bcm_cg_solve(self.h, logpath.encode())
        return self.get_loghyper()
