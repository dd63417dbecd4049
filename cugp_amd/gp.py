"""Python mirror of the reference's host interface over the C-ABI.

`Covsum` keeps the method names and argument meaning of the reference class
(cpp_serial_gp/covkernel.h:20-37); `BCM` those of distributed_gp/BCM.h:15-26.  Every method is a
thin call into libcugp.so -- no arithmetic happens in Python.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import check, f64, ptr


class Covsum:
    """One GP expert on one GPU.  Covsum(n, d) as covkernel.cpp:14-37; X, y are given per call as in
    the reference and uploaded whenever their CONTENTS differ from what the GPU holds (the reference
    recomputes K from the arguments on every call; comparing n*d doubles is nothing beside an O(n^3)
    evaluation, and an in-place edit of y or a recycled array address can not go unnoticed)."""

    def __init__(self, n, d, device=0, npad_min=0):
        self.n, self.d, self.device = int(n), int(d), int(device)
        self._h = C.c_void_p()
        check(capi.lib().cugp_create_padded(self.n, self.d, self.device, int(npad_min), C.byref(self._h)))
        self._data_key = None

    # -- lifetime --
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            capi.lib().cugp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # -- data --
    def set_data(self, X, y):
        X, y = f64(X), f64(y)
        if X.shape != (self.n, self.d) or y.shape != (self.n,):
            raise ValueError("expected X %s and y %s" % ((self.n, self.d), (self.n,)))
        check(capi.lib().cugp_set_data(self._h, ptr(X), ptr(y)))
        self._data_key = (X.copy(), y.copy())

    def set_data_device(self, dX_ptr, dy_ptr):
        check(capi.lib().cugp_set_data_device(self._h, C.c_void_p(dX_ptr), C.c_void_p(dy_ptr)))
        self._data_key = None

    def _bind(self, X, y):
        if X is None:
            return
        X = f64(X)
        y = f64(y) if y is not None else (self._data_key[1] if self._data_key is not None else np.zeros(self.n))
        k = self._data_key
        if k is None or k[0].shape != X.shape or not np.array_equal(k[0], X) or not np.array_equal(k[1], y):
            self.set_data(X, y)

    # -- hyper-parameters --
    def set_loghyperparam(self, hp):
        check(capi.lib().cugp_set_loghyper(self._h, ptr(f64(hp))))

    set_loghyper_eigen = set_loghyperparam

    def get_loghyperparam(self):
        out = np.empty(3)
        check(capi.lib().cugp_get_loghyper(self._h, ptr(out)))
        return out

    def get_param_dim(self):
        return self.d            # covkernel.cpp:661-663 returns numdim

    # -- objective --
    def compute_loglikelihood(self, X=None, y=None):
        self._bind(X, y)
        ll = C.c_double()
        check(capi.lib().cugp_loglik(self._h, C.byref(ll)))
        return ll.value

    def compute_gradient_loghyperparam(self, X=None, y=None):
        self._bind(X, y)
        g = np.empty(3)
        check(capi.lib().cugp_grad(self._h, ptr(g)))
        return g

    def loglik_grad(self, X=None, y=None):
        self._bind(X, y)
        ll = C.c_double()
        g = np.empty(3)
        check(capi.lib().cugp_loglik_grad(self._h, C.byref(ll), ptr(g)))
        return ll.value, g

    def enqueue(self, want_grad=True):
        check(capi.lib().cugp_loglik_grad_enqueue(self._h, 1 if want_grad else 0))

    def fetch(self):
        ll = C.c_double()
        g = np.empty(3)
        check(capi.lib().cugp_loglik_grad_fetch(self._h, C.byref(ll), ptr(g)))
        return ll.value, g

    def last_quad_logdet(self):
        q, d = C.c_double(), C.c_double()
        check(capi.lib().cugp_last_quad_logdet(self._h, C.byref(q), C.byref(d)))
        return q.value, d.value

    # -- intermediates --
    def compute_K_train(self, X=None):
        """Covsum::compute_K_train(X, out): the full symmetric n x n covariance (labels are not used)."""
        self._bind(X, None)
        K = np.empty((self.n, self.n))
        check(capi.lib().cugp_compute_K_train(self._h, ptr(K)))
        return K

    def compute_squared_dist(self, c):
        S = np.empty((self.n, self.n))
        check(capi.lib().cugp_compute_squared_dist(self._h, float(c), ptr(S)))
        return S

    def compute_k_test(self, Xt):
        Xt = f64(Xt).reshape(-1, self.d)
        Ks = np.empty((Xt.shape[0], self.n))
        check(capi.lib().cugp_compute_k_test(self._h, ptr(Xt), Xt.shape[0], ptr(Ks)))
        return Ks

    def get_cholesky(self):
        L = np.empty((self.n, self.n))
        check(capi.lib().cugp_get_cholesky(self._h, ptr(L)))
        return L

    def get_K_inverse(self):
        Ki = np.empty((self.n, self.n))
        check(capi.lib().cugp_get_K_inverse(self._h, ptr(Ki)))
        return Ki

    def get_alpha(self):
        a = np.empty(self.n)
        check(capi.lib().cugp_get_alpha(self._h, ptr(a)))
        return a

    # -- prediction --
    def compute_test_means_and_variances(self, X, y, Xtest):
        self._bind(X, y)
        Xt = f64(Xtest).reshape(-1, self.d)
        m, v = np.empty(Xt.shape[0]), np.empty(Xt.shape[0])
        check(capi.lib().cugp_predict(self._h, ptr(Xt), Xt.shape[0], ptr(m), ptr(v)))
        return m, v

    @staticmethod
    def get_negative_log_predprob(actual, predmean, predvar):
        a, m, v = f64(actual), f64(predmean), f64(predvar)
        out = C.c_double()
        check(capi.lib().cugp_nlpp(ptr(a), ptr(m), ptr(v), a.shape[0], C.byref(out)))
        return out.value

    # -- optimisers --
    def cg_solve(self, X=None, y=None, budget=100):
        """Covsum::cg_solve; returns the evaluation trace [n_evals, 4] = (hp0, hp1, hp2, -LL)."""
        self._bind(X, y)
        tr = np.zeros((4 * budget + 8, 4))
        ne = C.c_int()
        check(capi.lib().cugp_cg_solve(self._h, budget, ptr(tr), tr.shape[0], C.byref(ne)))
        return tr[: ne.value]

    def cg_solve_sparing(self, X=None, y=None, budget=100):
        """Opt-in: the same line search, but the gradient (two thirds of an evaluation) only where the search reads
        it.  -> (trace, gradient evaluations made)."""
        self._bind(X, y)
        tr = np.zeros((4 * budget + 8, 4))
        ne, ng = C.c_int(), C.c_int()
        check(capi.lib().cugp_cg_solve_sparing(self._h, budget, ptr(tr), tr.shape[0], C.byref(ne), C.byref(ng)))
        return tr[: ne.value], ng.value

    def rprop_solve(self, X=None, y=None, iters=100):
        self._bind(X, y)
        tr = np.zeros((2 * iters + 8, 4))
        ne = C.c_int()
        check(capi.lib().cugp_rprop_solve(self._h, iters, ptr(tr), tr.shape[0], C.byref(ne)))
        return tr[: ne.value]

    # -- timing --
    def set_profiling(self, level):
        check(capi.lib().cugp_set_profiling(self._h, int(level)))

    def set_overlap(self, enable):
        """Inverse blocks on further streams while the factorisation runs (default on; cugp_set_overlap)."""
        check(capi.lib().cugp_set_overlap(self._h, 1 if enable else 0))

    def set_tuning(self, key, value, own=True):
        """One launch-shape key (kernels.h TUNE_*) for THIS handle alone (cugp_set_handle_tuning); own=False hands the
        key back to the process default (cugp_set_tuning)."""
        check(capi.lib().cugp_set_handle_tuning(self._h, int(key), int(value), 1 if own else 0))

    def get_tuning(self, key):
        v = C.c_int()
        check(capi.lib().cugp_get_handle_tuning(self._h, int(key), C.byref(v)))
        return v.value

    def phase_ms(self):
        """Main-stream phases of the last evaluation.  With the overlap on, "potrf" includes the inverse blocks
        running beside it and "trtri" is what was left of them when the factorisation ended ("lauum" ~ 0)."""
        ms = np.empty(6)
        check(capi.lib().cugp_get_phase_ms(self._h, ptr(ms)))
        return dict(zip(("kbuild", "potrf", "trtri", "lauum", "tail", "total"), ms.tolist()))

    def kernel_stats(self, reset=False, kind=0):
        """Per-launch timings of one kernel kind (include/cugp.h: cugp_get_kernel_stats_kind); "disp_ms" (profiling
        level 5): the same launches from the end of the launch in front of each on its stream."""
        s, n, f, dms = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
        check(capi.lib().cugp_get_kernel_stats_dispatch_ms(self._h, int(kind), C.byref(dms)))
        check(capi.lib().cugp_get_kernel_stats_kind(self._h, int(kind), C.byref(s), C.byref(n), C.byref(f),
                                                    1 if reset else 0))
        return {"sum_ms": s.value, "launches": n.value, "flop": f.value, "disp_ms": dms.value}


class Comm:
    """The library's own RCCL communicator (cugp_comm_*, csrc/comm.cpp): one process per GPU, expert k on rank k mod W.
    `unique_id`: the 128 bytes rank 0 got from Comm.unique_id(), handed to every rank by the caller (ShardedBCM
    broadcasts them through torch.distributed); None with world == 1: no communicator, nothing to exchange."""

    ID_BYTES = 128

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * Comm.ID_BYTES)()
        check(capi.lib().cugp_comm_unique_id(buf, Comm.ID_BYTES))
        return bytes(buf)

    def __init__(self, unique_id, rank, world, device):
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        self._h = C.c_void_p()
        idbuf = (C.c_ubyte * Comm.ID_BYTES).from_buffer_copy(unique_id) if unique_id is not None else None
        check(capi.lib().cugp_comm_create(idbuf, Comm.ID_BYTES if unique_id is not None else 0, self.rank, self.world,
                                          self.device, C.byref(self._h)))

    def loglik_grad_allgather(self, bcm, per):
        """One sharded objective evaluation: this rank's experts (`bcm`: a BCM, or None on a rank that owns none)
        evaluated, everybody's rows gathered -> [world * per, 4] (rank r's i-th expert in row r * per + i)."""
        out = np.empty((self.world * int(per), 4))
        check(capi.lib().cugp_bcm_loglik_grad_allgather(bcm._h if bcm is not None else None, self._h, int(per), ptr(out)))
        return out

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            capi.lib().cugp_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BCM:
    """Experts resident on the GPU(s) of this process (class BCM, distributed_gp/BCM.h).  `devices` lists the
    GPUs (expert k on devices[k mod len], cg_solver.cpp:93; default: the one `device`).  `BCM.split` reproduces
    the reference constructor's row partition (BCM.cpp:85-110)."""

    def __init__(self, rows, d, device=0, devices=None):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        devs = np.ascontiguousarray([device] if devices is None else list(devices), dtype=np.int32)
        self.rows, self.d, self.device, self.devices = rows.tolist(), int(d), int(devs[0]), devs.tolist()
        self._h = C.c_void_p()
        check(capi.lib().cugp_bcm_create_multi(len(self.devices), devs.ctypes.data_as(capi._ip), len(self.rows),
                                               rows.ctypes.data_as(capi._ip), self.d, C.byref(self._h)))

    @classmethod
    def split(cls, X, y, K, device=0, devices=None):
        X, y = f64(X), f64(y)
        N, D = X.shape
        part = N // K
        rows = [part] * (K - 1) + [N - part * (K - 1)]
        b = cls(rows, D, device, devices)
        off = 0
        for k in range(K):
            b.set_expert_data(k, X[off: off + rows[k]], y[off: off + rows[k]])
            off += part
        return b

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            capi.lib().cugp_bcm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_expert_data(self, k, X, y):
        X, y = f64(X), f64(y)
        check(capi.lib().cugp_bcm_set_expert_data(self._h, k, ptr(X), ptr(y)))

    def set_BCM_log_hyperparam(self, hp):
        check(capi.lib().cugp_bcm_set_loghyper(self._h, ptr(f64(hp))))

    set_BCM_loghyper_eigen = set_BCM_log_hyperparam

    def get_loghyperparam(self):
        out = np.empty(3)
        check(capi.lib().cugp_bcm_get_loghyper(self._h, ptr(out)))
        return out

    def loglik_grad(self):
        """-> (sum LL, sum grad[3], per-expert LL) over the experts of this GPU."""
        ll = C.c_double()
        g = np.empty(3)
        per = np.empty(len(self.rows))
        check(capi.lib().cugp_bcm_loglik_grad(self._h, C.byref(ll), ptr(g), ptr(per)))
        return ll.value, g, per

    def loglik_grad_rows(self):
        """-> [K, 4] rows (LL_k, gradient of -LL_k): what a multi-GPU BCM all-reduces."""
        rows = np.empty((len(self.rows), 4))
        check(capi.lib().cugp_bcm_loglik_grad_rows(self._h, ptr(rows)))
        return rows

    def loglik_grad_rows_device(self, dev_rows_ptr, slots):
        """Leave every local expert's (LL_k, gradient) row in DEVICE memory: row slots[k] of the [., 4] buffer at
        dev_rows_ptr (same GPU) -- the payload of an all-reduce that never touches the host."""
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        check(capi.lib().cugp_bcm_loglik_grad_rows_device(self._h, C.c_void_p(dev_rows_ptr),
                                                          slots.ctypes.data_as(capi._ip)))

    def expert(self, k):
        """Borrowed view of expert k as a Covsum-like object (prediction, intermediates); owned by the BCM."""
        h = C.c_void_p()
        check(capi.lib().cugp_bcm_expert(self._h, int(k), C.byref(h)))
        e = Covsum.__new__(Covsum)
        e.n, e.d, e.device, e._h, e._data_key = self.rows[k], self.d, self.devices[k % len(self.devices)], h, None
        e.close = lambda: None                    # not ours to destroy
        return e

    def get_BCM_loglikelihood(self):
        return self.loglik_grad()[0]

    def get_BCM_gradient_hyper(self):
        return self.loglik_grad()[1]

    def predict_partial(self, Xt):
        Xt = f64(Xt).reshape(-1, self.d)
        sp, spm = np.empty(Xt.shape[0]), np.empty(Xt.shape[0])
        check(capi.lib().cugp_bcm_predict_partial(self._h, ptr(Xt), Xt.shape[0], ptr(sp), ptr(spm)))
        return sp, spm

    def compute_BCM_test_means_and_var(self, Xt):
        Xt = f64(Xt).reshape(-1, self.d)
        m, v = np.empty(Xt.shape[0]), np.empty(Xt.shape[0])
        check(capi.lib().cugp_bcm_predict(self._h, ptr(Xt), Xt.shape[0], ptr(m), ptr(v)))
        return m, v

    get_BCM_negative_log_predprob = staticmethod(Covsum.get_negative_log_predprob)

    def cg_solve(self, budget=100):
        tr = np.zeros((4 * budget + 8, 4))
        ne = C.c_int()
        check(capi.lib().cugp_bcm_cg_solve(self._h, budget, ptr(tr), tr.shape[0], C.byref(ne)))
        return tr[: ne.value]


def poe_finish(sum_prec, sum_prec_mean):
    sp, spm = f64(sum_prec), f64(sum_prec_mean)
    m, v = np.empty_like(sp), np.empty_like(sp)
    check(capi.lib().cugp_poe_finish(ptr(sp), ptr(spm), sp.shape[0], ptr(m), ptr(v)))
    return m, v


def cg_minimize(fn, theta, budget=100):
    """Host CG loop of the library on a Python objective fn(theta)->(f, g)."""
    def cb(_ctx, th, f, g):
        fv, gv = fn(np.array([th[0], th[1], th[2]]))
        f[0] = fv
        for i in range(3):
            g[i] = gv[i]
    th = f64(theta).copy()
    tr = np.zeros((4 * budget + 8, 4))
    ne = C.c_int()
    check(capi.lib().cugp_cg_minimize(capi.OBJECTIVE(cb), None, ptr(th), budget, ptr(tr), tr.shape[0],
                                      C.byref(ne)))
    return th, tr[: ne.value]


def cg_minimize_sparing(value_fn, gradient_fn, theta, budget=100):
    """Evaluation-sparing CG on Python callbacks value_fn(theta)->f, gradient_fn(theta)->g.
    -> (theta, trace, gradient evaluations made)."""
    def vf(_ctx, th, f):
        f[0] = value_fn(np.array([th[0], th[1], th[2]]))

    def gf(_ctx, th, g):
        gv = gradient_fn(np.array([th[0], th[1], th[2]]))
        for i in range(3):
            g[i] = gv[i]
    th = f64(theta).copy()
    tr = np.zeros((4 * budget + 8, 4))
    ne, ng = C.c_int(), C.c_int()
    check(capi.lib().cugp_cg_minimize_sparing(capi.VALUE_FN(vf), capi.GRADIENT_FN(gf), None, ptr(th), budget,
                                              ptr(tr), tr.shape[0], C.byref(ne), C.byref(ng)))
    return th, tr[: ne.value], ng.value


def rprop_minimize(fn, theta, iters=100):
    def cb(_ctx, th, f, g):
        fv, gv = fn(np.array([th[0], th[1], th[2]]))
        f[0] = fv
        for i in range(3):
            g[i] = gv[i]
    th = f64(theta).copy()
    tr = np.zeros((2 * iters + 8, 4))
    ne = C.c_int()
    check(capi.lib().cugp_rprop_minimize(capi.OBJECTIVE(cb), None, ptr(th), iters, ptr(tr), tr.shape[0],
                                         C.byref(ne)))
    return th, tr[: ne.value]


def test_gemm_nt(A, B, device=0):
    A, B = f64(A), f64(B)
    m, k = A.shape
    n = B.shape[0]
    Cm = np.empty((m, n))
    check(capi.lib().cugp_test_gemm_nt(m, n, k, ptr(A), ptr(B), ptr(Cm), device))
    return Cm


def mfma_peak_tflops(device=0):
    out = C.c_double()
    check(capi.lib().cugp_mfma_peak_tflops(device, C.byref(out)))
    return out.value


def potrf(K, device=0):
    K = f64(K)
    L = np.empty_like(K)
    check(capi.lib().cugp_potrf(K.shape[0], ptr(K), ptr(L), device))
    return L


def potri(K, device=0):
    K = f64(K)
    Ki = np.empty_like(K)
    check(capi.lib().cugp_potri(K.shape[0], ptr(K), ptr(Ki), device))
    return Ki


def chol_and_det(K, y, device=0):
    K, y = f64(K), f64(y)
    q, d = C.c_double(), C.c_double()
    check(capi.lib().cugp_chol_and_det(K.shape[0], ptr(K), ptr(y), C.byref(q), C.byref(d), device))
    return q.value, d.value


def potrs_vec(K, y, device=0):
    K, y = f64(K), f64(y)
    x = np.empty(K.shape[0])
    check(capi.lib().cugp_potrs_vec(K.shape[0], ptr(K), ptr(y), ptr(x), device))
    return x
