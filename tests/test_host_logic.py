"""Host logic of the product library that needs no GPU: the optimisers (minimize.cpp) against the
oracle's restatement on identical callbacks, the scalar prediction helpers, the row partition and the
ctypes surface.  The library is LOADED here (hipcc cross-compiled it) but no device call is made."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import cugp_amd.gp as gp
from cugp_amd import capi
from cugp_amd.bcm import expert_owner, split_rows
from conftest import HP_BCM, ROOT


def test_header_symbols_all_exported_and_bound():
    text = open(os.path.join(ROOT, "include", "cugp.h")).read()
    declared = set(re.findall(r"\b(cugp_[A-Za-z0-9_]+)\s*\(", text))
    declared -= {"cugp_objective_fn", "cugp_value_fn", "cugp_gradient_fn"}
    lib = capi.lib()
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert declared == set(capi.SIGNATURES), declared ^ set(capi.SIGNATURES)
    assert lib.cugp_version() >= 100


def test_one_hip_runtime_per_process_whatever_the_import_order():
    """capi.lib() before `import torch`: the process must end up with ONE libamdhip64 (torch's copy when a torch wheel
    is installed -- capi._share_torch_hip_runtime), and torch must still import.  Fresh interpreter: this one has
    imported both long ago."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from cugp_amd import capi\n"
            "capi.lib()\n"
            "import torch\n"
            "libs = sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l))\n"
            "print('RUNTIMES', len(libs), libs)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RUNTIMES 1 " in r.stdout, r.stdout


def test_no_device_is_an_error_not_a_fallback():
    """Without a GPU the product path must fail loudly (error code + message), never compute on the CPU."""
    lib = capi.lib()
    n = C.c_int(-1)
    rc = lib.cugp_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible")
    h = C.c_void_p()
    rc = lib.cugp_create(16, 2, 0, C.byref(h))
    assert rc == -4 and not h.value
    assert b"no HIP device" in lib.cugp_last_error()
    with pytest.raises(capi.CugpError):
        gp.Covsum(16, 2)
    with pytest.raises(capi.CugpError):
        gp.potrf(np.eye(4))
    rc = lib.cugp_create_padded(16, 2, 0, 256, C.byref(h))
    assert rc == -4 and not h.value
    with pytest.raises(capi.CugpError):                  # the BCM layer creates its experts the same way
        gp.BCM([8, 8], 2)


def test_argument_validation():
    lib = capi.lib()
    assert lib.cugp_create(0, 2, 0, C.byref(C.c_void_p())) == -1
    assert lib.cugp_cg_minimize(capi.OBJECTIVE(lambda *a: None), None, None, 10, None, 0, None) == -1
    assert lib.cugp_destroy(None) == 0 and lib.cugp_bcm_destroy(None) == 0
    out = C.c_double()
    assert lib.cugp_nlpp(None, None, None, 3, C.byref(out)) == -1


def _oracle_objective(oracle, X, y):
    def fn(th):
        return -oracle.loglik(X, y, th), oracle.grad(X, y, th)
    return fn


def test_cg_minimize_bitwise_vs_oracle(oracle, si128):
    """Same callback, same start: the library's CG loop and the oracle's must take identical steps."""
    X, y = si128
    fn = _oracle_objective(oracle, X[:48], y[:48])
    th_lib, tr_lib = gp.cg_minimize(fn, HP_BCM, 60)
    th_or, tr_or = oracle.cg_minimize(fn, HP_BCM, 60)
    assert tr_lib.shape == tr_or.shape
    assert np.array_equal(tr_lib, tr_or)
    assert np.array_equal(th_lib, th_or)


def test_cg_minimize_reproduces_reference_run(oracle, si128, golden_si128):
    X, y = si128
    c = golden_si128["cg"][0]
    th, tr = gp.cg_minimize(_oracle_objective(oracle, X, y), c["hp0"], 100)
    assert np.allclose(th, c["final_hp"], rtol=0, atol=1e-12)
    probes = np.array([p[1:] for p in c["please_see"] if p[0] in (1, 2)])
    assert np.allclose(tr[1:, :3], probes, atol=6e-7)


def test_cg_minimize_nan_bisects(oracle):
    """A probe that returns NaN (non-PD covariance) halves the step instead of aborting
    (covkernel.cpp:509-524): the probe after the NaN one is the midpoint (x2 + x3) / 2 with x2 = 0."""
    def make():
        count = [0]

        def fn(th):
            count[0] += 1
            if count[0] == 2:
                return float("nan"), np.array([np.nan, 0.0, 0.0])
            f = (th[0] - 0.5) ** 2 + 2 * (th[1] + 1) ** 2 + 0.5 * th[2] ** 2
            return f, np.array([2 * (th[0] - 0.5), 4 * (th[1] + 1), th[2]])
        return fn
    start = np.array([-3.0, 2.0, 1.0])
    th, tr = gp.cg_minimize(make(), start, 40)
    assert np.isnan(tr[1, 3]) and np.isfinite(tr[2, 3])
    assert np.allclose(tr[2, :3] - start, 0.5 * (tr[1, :3] - start), rtol=1e-12, atol=1e-15)
    assert np.isfinite(tr[-1, 3]) and np.allclose(th, [0.5, -1.0, 0.0], atol=1e-4)
    th2, tr2 = oracle.cg_minimize(make(), start, 40)
    assert np.array_equal(tr[~np.isnan(tr[:, 3])], tr2[~np.isnan(tr2[:, 3])]) and np.array_equal(th, th2)


def test_rprop_minimize_bitwise_vs_oracle(oracle, si128):
    X, y = si128
    fn = _oracle_objective(oracle, X[:40], y[:40])
    th_lib, tr_lib = gp.rprop_minimize(fn, HP_BCM, 25)
    th_or, tr_or = oracle.rprop_minimize(fn, HP_BCM, 25)
    assert np.array_equal(tr_lib, tr_or) and np.array_equal(th_lib, th_or)


def test_nlpp_and_poe_vs_oracle(oracle):
    rng = np.random.default_rng(1)
    a, m, v = rng.standard_normal(9), rng.standard_normal(9), rng.uniform(0.1, 3, 9)
    assert gp.Covsum.get_negative_log_predprob(a, m, v) == oracle.nlpp(a, m, v)
    means, vars_ = rng.standard_normal((5, 9)), rng.uniform(0.1, 3, (5, 9))
    sp = np.zeros(9)
    spm = np.zeros(9)
    for e in range(5):                      # BCM.cpp:51-55 accumulation order
        inv = 1.0 / vars_[e]
        sp += inv
        spm += inv * means[e]
    pm, pv = gp.poe_finish(sp, spm)
    om, ov = oracle.poe(means, vars_)
    assert np.array_equal(pm, om) and np.array_equal(pv, ov)


def test_row_partition_matches_reference(oracle, si128):
    X, y = si128
    for K in (1, 3, 4, 7):
        b = oracle.bcm(X, y, K, HP_BCM)
        assert split_rows(128, K) == [b.expert_rows(k) for k in range(K)]
    assert [expert_owner(k, 3) for k in range(7)] == [0, 1, 2, 0, 1, 2, 0]     # chunk i -> worker i mod W


# ---------------------------------------------------------------------------------------------
# two-speed Cholesky: the launch schedule (cugp_potrf_plan = the arithmetic enqueue_potrf uses)
# ---------------------------------------------------------------------------------------------


def _replay_potrf_plan(nt, P, near, S=1):
    """Replay the schedule on a set model.  applied[(i, j)] = k tiles subtracted from tile (i, j) so far, in
    launch order (one in-order stream).  Checks that every tile has seen exactly k = 0..j-1, ascending, when its
    column is solved / its diagonal block factored, and that the far boundary never moves backwards."""
    lib = capi.lib()
    applied = {(i, j): [] for j in range(nt) for i in range(j, nt)}
    last_far = 0
    for kb in range(nt - 1):
        out = (C.c_int * 6)()
        assert lib.cugp_potrf_plan_sub(nt, P, near, S, kb, out) == 0
        k0, kw, a0, a1, wcol, ks = list(out)
        if S == 1:
            out5 = (C.c_int * 5)()
            assert lib.cugp_potrf_plan(nt, P, near, kb, out5) == 0 and list(out5) == [k0, kw, a0, a1, wcol] and ks == kb
        assert 0 <= ks <= kb and kb - ks < 4                         # at most SUBPANEL_MAX k tiles per pass
        # panel solve of column kb: every tile of the column is up to date
        for i in range(kb, nt):
            assert applied[(i, kb)] == list(range(kb)), (nt, P, near, kb, i, applied[(i, kb)])
        assert wcol >= 1
        far = kb + 1 + wcol                                          # first column outside this launch's window
        assert far <= nt
        if a1 > a0:
            p = kb // P
            assert (k0, kw) == (p * P, P) and k0 + kw == kb + 1      # the panel just completed ...
            assert a0 >= far and a1 == nt                            # ... goes to everything beyond the window
            assert a0 == far or wcol == 1                            # (a left-looking sub-panel step touches one column)
            assert a0 >= last_far                                    # the far boundary never moves backwards
            last_far = a0
            for j in range(a0, a1):
                for i in range(j, nt):
                    applied[(i, j)] += list(range(k0, k0 + kw))
        else:
            assert far == nt or kb % P != P - 1 or P == 1           # (P = 1 has no wide passes: the window is the whole trailing matrix)
        for j in range(kb + 1, far):
            for i in range(j, nt):
                applied[(i, j)] += list(range(ks, kb + 1))
        # the diagonal block factored inside this launch
        assert applied[(kb + 1, kb + 1)] == list(range(kb + 1)), (nt, P, near, kb, applied[(kb + 1, kb + 1)])
    for (i, j), ks in applied.items():
        assert ks == list(range(j)), (nt, P, near, i, j, ks)          # exactly once each, ascending


def test_potrf_plan_covers_every_update_exactly_once_in_order():
    for P in (1, 2, 3, 4, 5, 8, 16, 32):                     # 16 / 500 is the shipped default
        for near in (1, 40, 300, 500, 700, 5000):
            for nt in list(range(2, 30)) + [40, 63, 64, 79]:
                _replay_potrf_plan(nt, P, near)
                for S in (2, 4):                             # sub-panelled near window (tuning key 17)
                    _replay_potrf_plan(nt, P, near, S)


def test_cg_sparing_takes_the_default_trajectory_with_fewer_gradients(oracle, si128):
    """Opt-in evaluation-sparing CG: with an objective whose value does not depend on which half is called, every
    probe point and value equals the default loop's (covkernel.cpp:405-647), and probes above the line search's
    starting value cost no gradient."""
    X, y = si128
    Xs, ys = X[:48], y[:48]
    calls = {"g": 0}

    def value(th):
        return -oracle.loglik(Xs, ys, th)

    def gradient(th):
        calls["g"] += 1
        return oracle.grad(Xs, ys, th)

    th0, tr0 = gp.cg_minimize(lambda th: (value(th), oracle.grad(Xs, ys, th)), HP_BCM, 60)
    th1, tr1, ng = gp.cg_minimize_sparing(value, gradient, HP_BCM, 60)
    assert np.array_equal(tr0, tr1) and np.array_equal(th0, th1)
    assert ng == calls["g"] and ng <= tr1.shape[0]
    print("sparing CG: %d probes, %d gradients" % (tr1.shape[0], ng))
    assert ng < tr1.shape[0]                          # this run has rejected probes (f above the starting value)
