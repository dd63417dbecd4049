"""Guarantees about handles that used to live in tools/ only (round 5): many handles in flight together, every hand-over
block size, the bounded stage barrier of k_trtri_block running out, per-handle tuning.  The reference is one GP per
process (file-scope globals, cuda_scalingdist/main.cpp:14-67); the C-ABI promises thread-compatible handles
(INTEGRATION.md, section E), and these tests hold it to that on the GPU."""
import threading

import numpy as np
import pytest

from conftest import synth

pytestmark = pytest.mark.gpu

HP = np.array([np.log(3.0), 0.0, np.log(0.1)])
TUNE_PIPE_BLOCK, TUNE_BARRIER_SPIN, TUNE_ZFUSE = 3, 16, 18


@pytest.fixture(scope="module")
def gp_mod():
    import cugp_amd.gp as gp
    return gp


def make(gp_mod, n, d, seed):
    X, y = synth(n, d=d, seed=seed)
    g = gp_mod.Covsum(n, d)
    g.set_data(X, y)
    return g


def test_sixteen_handles_in_flight_equal_the_evaluations_alone(gp_mod):
    """16 handles enqueue before the first fetch: every handle forks its inverse blocks to three further streams, so
    dozens of k_trtri_block grids (stage barriers) are in flight beside hundreds of tile workgroups.  Every result must
    be the bits of the same handle evaluated alone (tools/concurrent_handles.py, shrunk)."""
    sizes = [1100, 1500, 900, 1300, 700, 1700, 515, 2100]
    hs = [make(gp_mod, sizes[i % len(sizes)], 5, 100 + i) for i in range(16)]
    alone = []
    for g in hs:
        g.set_loghyperparam(HP)
        alone.append(g.loglik_grad())
    for r in range(2):
        for g in hs:
            g.set_loghyperparam(HP + 1e-3 * (r + 1))
            g.enqueue(True)
        for g in hs:
            g.fetch()
        for g in hs:
            g.set_loghyperparam(HP)
            g.enqueue(True)
        for i, g in enumerate(hs):
            ll, gr = g.fetch()
            assert ll == alone[i][0] and tuple(gr) == tuple(alone[i][1]), (i, ll, alone[i])
    for g in hs:
        g.close()


def test_twelve_handles_with_whole_matrix_barrier_grids(gp_mod):
    """The case the barrier budget exists for (cugp_capi.cpp: barrier_cap): overlap off, 9..16 tiles -- every handle's
    whole inverse is ONE k_trtri_block launch that would like 64 co-resident workgroups; 12 of them in flight together
    want 768 of the 512 workgroup slots.  With the budget each takes 384 / 12 = 32: nothing stalls, nothing times out,
    and the bits are those of the handle alone (where the launch takes all 64: the workgroup count changes no result)."""
    rows = [1153, 1300, 1500, 1700, 1900, 2048, 1200, 1400, 1600, 1800, 2000, 1250]      # 10..16 tiles
    solo = make(gp_mod, rows[0], 4, 300)
    solo.set_overlap(False)
    solo.set_loghyperparam(HP)
    first_alone = solo.loglik_grad()                     # one live handle: its launch holds 64 workgroups
    solo.close()
    hs = [make(gp_mod, n, 4, 300 + i) for i, n in enumerate(rows)]
    for g in hs:
        g.set_overlap(False)
    for r in range(3):
        for g in hs:
            g.set_loghyperparam(HP + 1e-3 * r)
            g.enqueue(True)
        res = [g.fetch() for g in hs]                    # (a stage wait that ran out raises CUGP_ERR_DEVICE here)
        assert all(np.isfinite(ll) and np.all(np.isfinite(gr)) for ll, gr in res)
        if r == 0:
            assert res[0][0] == first_alone[0] and tuple(res[0][1]) == tuple(first_alone[1])
            again = []
            for g in hs:                                 # one at a time, same budget share
                g.set_loghyperparam(HP + 1.0)
                g.loglik_grad()
                g.set_loghyperparam(HP)
                again.append(g.loglik_grad())
            for a, b in zip(res, again):
                assert a[0] == b[0] and tuple(a[1]) == tuple(b[1])
    for g in hs:
        g.close()


def test_more_handles_than_the_barrier_budget_serves(gp_mod):
    """A handle that asks for its share of the barrier budget while more than 24 handles are alive on the device gets
    none: its block inverses go launch by launch (k_trtri_diag + k_trtri_level: nothing waits for anything).  Same tile
    code in the same order: the same bits as the one-launch form of a handle that holds a share."""
    g = make(gp_mod, 1500, 4, 77)
    g.set_loghyperparam(HP)
    one_launch = g.loglik_grad()                         # alone on the device: 64 barrier workgroups reserved
    crowd = [gp_mod.Covsum(130, 4) for _ in range(25)]   # 26 live handles now
    late = make(gp_mod, 1500, 4, 77)                     # the same data in a handle that comes too late for a share
    late.set_loghyperparam(HP)
    by_launches = late.loglik_grad()
    for c in crowd:
        c.close()
    assert one_launch[0] == by_launches[0] and tuple(one_launch[1]) == tuple(by_launches[1])
    g.close()
    late.close()


def test_graphs_captured_before_more_handles_arrive(gp_mod):
    """a captured graph bakes its k_trtri_block grid in.  Six overlap-off handles capture their graphs
    with 64 barrier workgroups each (the whole pool of 384); six more handles created afterwards must not push the sum
    over the 512 workgroup slots -- they get what is left of the pool (nothing: launch by launch), the first six keep their
    reserved grids, and all twelve in flight together finish with the bits each gets alone."""
    rows = [1153, 1300, 1500, 1700, 1900, 2048, 1200, 1400, 1600, 1800, 2000, 1250]      # 10..16 tiles: graph replays
    first = [make(gp_mod, n, 4, 500 + i) for i, n in enumerate(rows[:6])]
    for g in first:
        g.set_overlap(False)
        g.set_loghyperparam(HP)
    alone = [g.loglik_grad() for g in first]             # captures each handle's graph
    second = [make(gp_mod, n, 4, 506 + i) for i, n in enumerate(rows[6:])]
    for g in second:
        g.set_overlap(False)
        g.set_loghyperparam(HP)
    alone += [g.loglik_grad() for g in second]
    hs = first + second
    for r in range(3):
        for g in hs:
            g.set_loghyperparam(HP + (1e-3 * r if r else 0.0))
            g.enqueue(True)
        res = [g.fetch() for g in hs]                    # (a stage wait that ran out raises CUGP_ERR_DEVICE here)
        assert all(np.isfinite(ll) and np.all(np.isfinite(gr)) for ll, gr in res)
        if r == 0:
            for a, b in zip(res, alone):
                assert a[0] == b[0] and tuple(a[1]) == tuple(b[1])
    for g in hs:
        g.close()


@pytest.mark.parametrize("n", [130, 515, 900, 1500, 2100, 2500])
def test_every_hand_over_block_size(gp_mod, n):
    """Hand-over blocks of 1..16 tiles (TUNE_PIPE_BLOCK, set for this handle only) against the single-stream evaluation
    of the same handle: LL 1e-12, gradient 1e-9, K^-1 1e-11 of its largest entry (the partition changes summation
    orders, nothing else) -- tools/block_sweep.py, shrunk."""
    g = make(gp_mod, n, 7, n)
    g.set_loghyperparam(HP)
    g.set_overlap(False)
    ll0, g0 = g.loglik_grad()
    K0 = g.get_K_inverse()
    g.set_overlap(True)
    for w in (1, 2, 3, 4, 5, 7, 8, 11, 16):
        g.set_tuning(TUNE_PIPE_BLOCK, w)
        assert g.get_tuning(TUNE_PIPE_BLOCK) == w
        g.set_loghyperparam(HP + 1e-9)
        g.loglik_grad()
        g.set_loghyperparam(HP)
        ll, gr = g.loglik_grad()
        Ki = g.get_K_inverse()
        assert np.isfinite(ll) and abs(ll - ll0) <= 1e-12 * max(1.0, abs(ll0)), (w, ll, ll0)
        assert np.max(np.abs(gr - g0) / (np.abs(g0) + 1e-9 * np.max(np.abs(g0)))) < 1e-9, (w, gr, g0)
        assert np.max(np.abs(Ki - K0)) <= 1e-11 * np.max(np.abs(K0)), w
    g.close()


def test_a_stage_barrier_that_runs_out_is_an_error_not_a_hang(gp_mod):
    """TUNE_BARRIER_SPIN = 0 on one handle: every workgroup of its k_trtri_block launches gives up at its first stage
    barrier without waiting.  The evaluation must come back (no hung device) as CUGP_ERR_DEVICE, keep nothing (no
    'valid' factor with a poisoned log-determinant share), and the next evaluation on the SAME handle -- spin bound
    restored -- must be clean and equal to an untouched handle's.  A second handle is never affected."""
    from cugp_amd import capi
    n = 1500
    g, other = make(gp_mod, n, 6, 5), make(gp_mod, n, 6, 5)
    g.set_loghyperparam(HP)
    other.set_loghyperparam(HP)
    want = other.loglik_grad()
    g.set_tuning(TUNE_BARRIER_SPIN, 0)
    with pytest.raises(capi.CugpError) as ei:
        g.loglik_grad()
    assert ei.value.code == capi.CUGP_ERR_DEVICE and "barrier" in str(ei.value)
    with pytest.raises(capi.CugpError):
        g.last_quad_logdet()                              # nothing of the abandoned evaluation is "available"
    assert other.get_tuning(TUNE_BARRIER_SPIN) == 1 << 21
    ll_o, g_o = other.loglik_grad()
    assert ll_o == want[0]
    g.set_tuning(TUNE_BARRIER_SPIN, 0, own=False)        # back to the process default
    assert g.get_tuning(TUNE_BARRIER_SPIN) == 1 << 21
    ll, gr = g.loglik_grad()
    assert ll == want[0] and tuple(gr) == tuple(want[1])
    # the single-stream continuation (cugp_loglik, then the gradient from the valid factor) goes through the same launch
    g.set_loghyperparam(HP + 0.25)
    g.compute_loglikelihood()
    g.set_tuning(TUNE_BARRIER_SPIN, 0)
    with pytest.raises(capi.CugpError):
        g.compute_gradient_loghyperparam()
    g.set_tuning(TUNE_BARRIER_SPIN, 0, own=False)
    other.set_loghyperparam(HP + 0.25)
    want2 = other.loglik_grad()
    ll2, gr2 = g.loglik_grad()                            # from the covariance build again
    assert abs(ll2 - want2[0]) <= 1e-12 * abs(want2[0]) and np.allclose(gr2, want2[1], rtol=1e-9)
    g.close()
    other.close()


def test_handle_tuning_from_two_threads(gp_mod):
    """Two threads, one handle each, different hand-over block sizes set per handle while the other thread evaluates: each
    handle's results equal its own single-threaded results at that setting (cugp_set_tuning was a shared mutable global
    until round 5)."""
    n = 1300
    hs = [make(gp_mod, n, 5, 11), make(gp_mod, n, 5, 11)]
    ws = [1, 3]
    want = []
    for g, w in zip(hs, ws):
        g.set_tuning(TUNE_PIPE_BLOCK, w)
        g.set_loghyperparam(HP)
        want.append(g.loglik_grad())
    errs = []

    def run(i):
        try:
            g = hs[i]
            for r in range(12):
                g.set_tuning(TUNE_PIPE_BLOCK, ws[i])
                g.set_loghyperparam(HP + 1e-3)
                g.loglik_grad()
                g.set_loghyperparam(HP)
                ll, gr = g.loglik_grad()
                assert ll == want[i][0] and tuple(gr) == tuple(want[i][1]), (i, r)
        except Exception as e:                           # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for g in hs:
        g.close()


def test_profiling_levels_time_the_same_launches_and_change_no_result(gp_mod):
    """cugp_set_profiling 3 (event pairs around the launches), 4 (the launches' own start / stop events) and 5 (the
    launches stamped by their own workgroups, nothing added to the streams): the same results as with profiling off, bit
    for bit, the same launches counted with the same algorithmic flop, durations of the same order; level 5 also times
    the covariance build (kind 10, its `flop` is bytes) and the panel solves (kind 11), and its dispatch-based duration
    of a step launch (from the end of the panel solve in front of it) is never below the workgroup-based one."""
    n = 2600                                              # 21 tiles: hand-over blocks, no two-speed panels
    g = make(gp_mod, n, 6, 3)
    g.set_loghyperparam(HP)
    want = g.loglik_grad()
    stats = {}
    for level in (3, 4, 5):
        g.set_profiling(level)
        g.set_loghyperparam(HP + 1e-3)
        g.loglik_grad()
        for kd in range(12):
            g.kernel_stats(reset=True, kind=kd)
        g.set_loghyperparam(HP)
        ll, gr = g.loglik_grad()
        assert ll == want[0] and tuple(gr) == tuple(want[1]), level
        stats[level] = {kd: g.kernel_stats(kind=kd) for kd in range(12)}
    g.set_profiling(0)
    for kd in (0, 2, 3, 4, 5, 8):                        # step launches, bordering, K^-1 shares, block inverses
        a, b, c = stats[3][kd], stats[4][kd], stats[5][kd]
        assert a["launches"] == b["launches"] == c["launches"], (kd, a, b, c)
        assert a["flop"] == b["flop"] == c["flop"], kd
        if a["launches"] > 0:
            assert 0.3 * a["sum_ms"] <= c["sum_ms"] <= 1.5 * a["sum_ms"], (kd, a, c)
            assert 0.5 * a["sum_ms"] <= b["sum_ms"] <= 1.5 * a["sum_ms"], (kd, a, b)
    assert stats[5][0]["launches"] == 20 and stats[5][11]["launches"] == 20      # 20 steps, 20 panel solves
    assert stats[5][0]["disp_ms"] >= stats[5][0]["sum_ms"] > 0.0
    kb = stats[5][10]
    assert kb["launches"] == 1 and kb["flop"] > 8.0 * n * n / 2 and 0.0 < kb["sum_ms"] < 1.0
    assert stats[3][10]["launches"] == 0 and stats[4][11]["launches"] == 0     # (levels 3, 4 time the MFMA kernels only)
    # prediction's product is timed from level 3 on (kind 9)
    g.set_profiling(5)
    X, _ = synth(64, d=6, seed=9)
    g.kernel_stats(reset=True, kind=9)
    g.compute_test_means_and_variances(None, None, X)
    p = g.kernel_stats(kind=9)
    assert p["launches"] == 1 and p["sum_ms"] > 0.0 and p["flop"] > 0.0
    g.close()


@pytest.mark.parametrize("n", [100, 130, 515, 1500, 2600, 4200])
def test_forward_substitution_inside_the_factorisation(gp_mod, oracle, n):
    """LL-only evaluations (Covsum::compute_loglikelihood, covkernel.cpp:118-129: Cholesky, then L z = y by
    matrixops.cpp:113-185's forward sweep): since round 5 z is computed inside the factorisation's own launches (one more
    workgroup per panel solve, nt - kb - 1 more per step launch; TUNE_ZFUSE = 1) instead of by 2 nt small launches behind
    it.  Same quadratic form and log-determinant as the launch-by-launch substitution to rounding (the diagonal blocks
    are applied as two 64x64 inverses instead of one 128x128 inverse), the same as the gradient path's z = L^-1 y, and the
    oracle's to 1e-8.  1 tile, 2 tiles, ragged, config-5 size, many hand-over blocks, the two-speed factorisation."""
    X, y = synth(n, d=6, seed=n)
    g = gp_mod.Covsum(n, 6)
    g.set_data(X, y)
    res = {}
    for mode in (1, 0):
        g.set_tuning(TUNE_ZFUSE, mode)
        g.set_loghyperparam(HP + 1e-3)
        g.compute_loglikelihood()
        g.set_loghyperparam(HP)
        ll = g.compute_loglikelihood()
        res[mode] = (ll,) + g.last_quad_logdet()
    assert res[1][2] == res[0][2]                                        # the same factor: the same log-determinant
    assert abs(res[1][1] - res[0][1]) <= 1e-12 * abs(res[0][1]), res     # y' K^-1 y
    assert abs(res[1][0] - res[0][0]) <= 1e-12 * abs(res[0][0]), res
    g.set_loghyperparam(HP + 1e-3)
    g.loglik_grad()
    g.set_loghyperparam(HP)
    ll_g, _ = g.loglik_grad()
    assert abs(ll_g - res[1][0]) <= 1e-11 * abs(ll_g), (ll_g, res[1][0])
    if n <= 1500:
        llo = oracle.loglik(X, y, HP)
        assert abs(res[1][0] - llo) <= 1e-8 * max(1.0, abs(llo)), (res[1][0], llo)
    # twice the same: bit-reproducible
    g.set_tuning(TUNE_ZFUSE, 1)
    g.set_loghyperparam(HP + 1e-3)
    g.compute_loglikelihood()
    g.set_loghyperparam(HP)
    assert g.compute_loglikelihood() == res[1][0]
    g.close()


def test_forward_substitution_in_a_group(gp_mod):
    """The same for experts that share launches (blockIdx.y = expert; group.h's internal entry points, which the BCM
    layer drives with the gradient on): an LL-only group evaluation, fused against the launch-by-launch substitution and
    against every expert evaluated alone."""
    import ctypes as C
    from cugp_amd import capi
    L = capi.lib()
    X, y = synth(4 * 700 + 11, d=5, seed=8)
    b = gp_mod.BCM.split(X, y, 4)
    b.set_BCM_log_hyperparam(HP)
    hs = (C.c_void_p * 4)(*[b.expert(k)._h for k in range(4)])
    grp = C.c_void_p()
    L.cugp_group_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
    L.cugp_group_eval.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.cugp_group_destroy.argtypes = [C.c_void_p]
    L.cugp_group_destroy.restype = None
    capi.check(L.cugp_group_create(hs, 4, C.byref(grp)))
    vals = {}
    try:
        for mode in (1, 0):
            capi.check(L.cugp_set_tuning(TUNE_ZFUSE, mode))
            ll = (C.c_double * 4)()
            b.set_BCM_log_hyperparam(HP + 1e-3)
            capi.check(L.cugp_group_eval(grp, 0, ll, None))
            b.set_BCM_log_hyperparam(HP)
            capi.check(L.cugp_group_eval(grp, 0, ll, None))
            vals[mode] = np.array(list(ll))
    finally:
        capi.check(L.cugp_set_tuning(TUNE_ZFUSE, 1))
        L.cugp_group_destroy(grp)
    assert np.all(np.abs(vals[1] - vals[0]) <= 1e-12 * np.abs(vals[0])), vals
    for k in range(4):
        e = b.expert(k)
        e.set_loghyperparam(HP + 1e-3)
        e.compute_loglikelihood()
        e.set_loghyperparam(HP)
        assert e.compute_loglikelihood() == vals[1][k], k               # alone (same padded size): the same bits
    b.close()


def test_bcm_prediction_right_after_new_hyperparameters(gp_mod, oracle):
    """BCM::compute_BCM_test_means_and_var (BCM.cpp:64-83) right after set_BCM_log_hyperparam (:123-130): the experts'
    inverse quantities are stale, and since round 5 ONE evaluation of the whole model brings them up to date (not one
    evaluation per expert inside cugp_predict); every expert's prediction is then in flight before the first is read.
    The same numbers as a prediction that follows an explicit evaluation, bit for bit, and the oracle's product of
    experts to 1e-8."""
    X, y = synth(5 * 300 + 17, d=6, seed=21)
    Xt = X[:33] * 0.5 + 0.1
    b = gp_mod.BCM.split(X, y, 5)
    b.set_BCM_log_hyperparam(HP)
    b.loglik_grad()
    m0, v0 = b.compute_BCM_test_means_and_var(Xt)
    b.set_BCM_log_hyperparam(HP + 0.3)
    b.compute_BCM_test_means_and_var(Xt)                     # stale -> grouped evaluation inside
    b.set_BCM_log_hyperparam(HP)
    m1, v1 = b.compute_BCM_test_means_and_var(Xt)            # stale again
    assert np.array_equal(m0, m1) and np.array_equal(v0, v1)
    mo, vo = oracle.bcm(X, y, 5, HP).predict(Xt)
    assert np.allclose(m1, mo, rtol=1e-8, atol=1e-8) and np.allclose(v1, vo, rtol=1e-8, atol=1e-8)
    b.close()
