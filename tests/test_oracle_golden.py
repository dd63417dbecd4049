"""Pins the CPU oracle (oracle/gp_oracle.c) to the reference: every function is checked against the
golden vectors that tests/golden/make_golden.py produced with the reference's OWN sources compiled
unmodified (oracle/Makefile -> oracle/_ref), and against the reference's committed run log
cuda_bettersinglenode_ver2/REF.  CPU only.  The restatement follows the reference operation for
operation (no FMA contraction), so agreement is expected to the last bit; tolerances of a few ulp are
left for libm differences across machines."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, HP_BCM

TIGHT = dict(rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_si128_loglik_grad_la(oracle, si128, golden_si128, idx):
    X, y = si128
    c = golden_si128["cases"][idx]
    hp = c["hp"]
    assert oracle.loglik(X, y, hp) == pytest.approx(c["ll"], rel=1e-14)
    assert np.allclose(oracle.grad(X, y, hp), c["grad"], **TIGHT)
    K = oracle.K_train(X, hp)
    assert np.allclose(K[5], c["K_row5"], **TIGHT) and np.allclose(np.diag(K), c["K_diag"], **TIGHT)
    L = oracle.cholesky(K)
    assert np.allclose(L[100], c["L_row100"], **TIGHT) and np.allclose(np.diag(L), c["L_diag"], **TIGHT)
    assert np.array_equal(np.triu(L, 1), np.zeros_like(L))
    Ki = oracle.K_inverse(K)
    assert np.allclose(Ki[7], c["Kinv_row7"], rtol=1e-12, atol=1e-12)
    assert np.trace(Ki) == pytest.approx(c["Kinv_trace"], rel=1e-13)
    q, ld = oracle.chol_and_det(K, y)
    assert q == pytest.approx(c["quad"], rel=1e-13) and ld == pytest.approx(c["logdet"], rel=1e-13)
    m, v = oracle.predict(X, y, hp, np.array(c["Xt"]))
    assert np.allclose(m, c["pred_mean"], **TIGHT) and np.allclose(v, c["pred_var"], **TIGHT)
    assert oracle.nlpp(y[:3], m, v) == pytest.approx(c["nlpp"], rel=1e-13)


def test_si128_matches_reference_log(oracle, si128, ref_log):
    """REF (the reference's committed stdout of the same 128-point run): LL and gradient at hp=1.5 and
    the final hyper-parameters, to the 6 decimals it prints."""
    X, y = si128
    assert round(oracle.loglik(X, y, HP_BCM), 6) == ref_log["loglik_values"][0]
    assert np.allclose(oracle.grad(X, y, HP_BCM), ref_log["gradients"][0], atol=5e-7)
    final, tr = oracle.cg_solve(X, y, HP_BCM)
    assert ref_log["please_see"][-1][0] == 3
    assert np.allclose(final, ref_log["please_see"][-1][1:], atol=2e-6)
    assert abs(-tr[-1, 3] - ref_log["loglik_values"][-1]) < 1e-5 or \
        abs(oracle.loglik(X, y, final) - ref_log["loglik_values"][-1]) < 5e-6
    # every probe the reference's GPU run printed ("PLEASE-SEE 1/2"), in order
    probes = np.array([p[1:] for p in ref_log["please_see"] if p[0] in (1, 2)])
    assert tr.shape[0] == probes.shape[0] + 1
    assert np.allclose(tr[1:, :3], probes, atol=5e-5)       # cuSOLVER path vs serial path: 5th digit


@pytest.mark.parametrize("idx", [0, 1])
def test_si128_cg_trace_vs_compiled_reference(oracle, si128, golden_si128, idx):
    X, y = si128
    c = golden_si128["cg"][idx]
    final, tr = oracle.cg_solve(X, y, c["hp0"])
    assert np.allclose(final, c["final_hp"], rtol=0, atol=1e-12)
    probes = np.array([p[1:] for p in c["please_see"] if p[0] in (1, 2)])
    assert tr.shape[0] == probes.shape[0] + 1
    assert np.allclose(tr[1:, :3], probes, atol=6e-7)       # the log prints 6 decimals
    assert oracle.loglik(X, y, final) == pytest.approx(c["final_ll"], rel=1e-12)


def test_si128_rprop(oracle, si128, golden_si128):
    X, y = si128
    c = golden_si128["rprop"]
    final, _ = oracle.rprop_solve(X, y, c["hp0"])
    assert np.allclose(final, c["final_hp"], rtol=0, atol=1e-12)


def test_si128_bcm(oracle, si128, golden_si128):
    X, y = si128
    c = golden_si128["bcm"]
    b = oracle.bcm(X, y, c["K"], c["hp"])
    ll, per = b.loglik()
    assert ll == pytest.approx(c["ll"], rel=1e-14)
    assert np.allclose(per, c["ll_per_expert_6dp"], atol=5e-7)
    assert np.allclose(b.grad(), c["grad"], **TIGHT)
    m, v = b.predict(np.array(c["Xt"]))
    assert np.allclose(m, c["pred_mean"], **TIGHT) and np.allclose(v, c["pred_var"], **TIGHT)
    u = c["uneven"]
    b3 = oracle.bcm(X, y, u["K"], u["hp"])
    assert [b3.expert_rows(k) for k in range(3)] == [(0, 42), (42, 42), (84, 44)]
    assert b3.loglik()[0] == pytest.approx(u["ll"], rel=1e-14)
    assert np.allclose(b3.grad(), u["grad"], **TIGHT)
    m, v = b3.predict(np.array(c["Xt"]))
    assert np.allclose(m, u["pred_mean"], **TIGHT) and np.allclose(v, u["pred_var"], **TIGHT)
    b4 = oracle.bcm(X, y, 4, c["cg"]["hp0"])
    final, _ = b4.cg_solve()
    assert np.allclose(final, c["cg"]["final_hp"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_sine_golden(oracle, sine, golden_sine, idx):
    """sine_dataset first 256 rows (two hyper-parameter points) and 1024 rows."""
    Xq, yq = sine
    c = golden_sine["cases"][idx]
    n = c["n"]
    X, y = Xq[:n], yq[:n]
    assert oracle.loglik(X, y, c["hp"]) == pytest.approx(c["ll"], rel=1e-14)
    g = oracle.grad(X, y, c["hp"])
    assert np.allclose(g, c["grad"], rtol=1e-12, atol=1e-12)
    if "pred_mean" in c:
        a, b = c["test_rows"]
        m, v = oracle.predict(X, y, c["hp"], Xq[a:b])
        assert np.allclose(m, c["pred_mean"], **TIGHT) and np.allclose(v, c["pred_var"], **TIGHT)
        assert oracle.nlpp(yq[a:b], m, v) == pytest.approx(c["nlpp"], rel=1e-13)


def test_sine_cg256(oracle, sine, golden_sine):
    Xq, yq = sine
    c = golden_sine["cg256"]
    final, tr = oracle.cg_solve(Xq[:256], yq[:256], c["hp0"])
    assert np.allclose(final, c["final_hp"], rtol=0, atol=1e-11)
    probes = np.array([p[1:] for p in c["please_see"] if p[0] in (1, 2)])
    assert tr.shape[0] == probes.shape[0] + 1


def test_degenerate_inputs(oracle):
    """n = 1, duplicated rows (singular without noise), non-PD -> NaN."""
    X = np.array([[0.3, -1.0]])
    y = np.array([0.7])
    hp = [0.1, 0.2, -0.5]
    k = np.exp(0.4) + np.exp(-1.0)
    assert oracle.loglik(X, y, hp) == pytest.approx(-0.5 * (0.49 / k + np.log(k) + 1.83787), rel=1e-14)
    X2 = np.zeros((4, 2))
    ll = oracle.loglik(X2, np.ones(4), [0.0, 0.0, -30.0])
    assert not np.isfinite(ll) or abs(ll) > 0     # singular to working precision: NaN / inf / huge, never a crash
    Kbad = np.eye(3)
    Kbad[2, 2] = -1.0
    q, ld = oracle.chol_and_det(Kbad, np.ones(3))
    assert np.isnan(ld)


def test_oracle_on_si24000_shards(oracle):
    """Config 5's data through the CPU restatement: two of the 16 shards of si24000 against the per-expert
    log-likelihoods the reference's BCM printed (6 decimals), and the dense-hp gradient contribution stays finite.
    (The full 16-shard sum and the 6000-/8192-/10000-row cases are GPU tests: the oracle needs minutes to hours there.)"""
    p = os.path.join(GOLDEN, "golden_r2", "si24000_bcm16.json")
    assert os.path.exists(p), "golden_r2/si24000_bcm16.json is missing (a committed fixture)"
    c = json.load(open(p))["cases"][0]
    d = np.load(os.path.join(GOLDEN, "data_si24000.npz"))
    for k in (0, 15):
        X = np.ascontiguousarray(d["X"][1500 * k:1500 * (k + 1)])
        y = np.ascontiguousarray(d["y"][1500 * k:1500 * (k + 1)])
        ll = oracle.loglik(X, y, c["hp"])
        assert abs(ll - c["ll_per_expert_6dp"][k]) <= 6e-7, (k, ll, c["ll_per_expert_6dp"][k])


def test_every_listed_golden_job_is_committed():
    """Every reference job tests/golden/run_jobs.sh runs (make_golden.py JOBS / JOBS_R3 / JOBS_R4) has its fixture in the
    tree, and so have the big single-matrix cases: the GPU tests that read them FAIL on a missing file, and this test says
    so on the CPU already (a deleted golden must not read as green)."""
    import ast
    src = open(os.path.join(GOLDEN, "make_golden.py")).read()
    names = []
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.Assign) and any(getattr(t, "id", "") in ("JOBS", "JOBS_R3", "JOBS_R4") for t in node.targets):
            names += list(eval(compile(ast.Expression(node.value), "make_golden.py", "eval"), {"range": range}))   # (plain list arithmetic)
    assert len(names) >= 30
    missing = [n for n in names if not os.path.exists(os.path.join(GOLDEN, "golden_r2", n + ".json"))]
    assert not missing, missing
    for f in ("golden_big_4096.json", "golden_big_8192.json", "data_siproper_9192.npz", "data_si24000.npz"):
        assert os.path.exists(os.path.join(GOLDEN, f)), f
