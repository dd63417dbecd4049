"""The C++ drop-in layer (cugp_amd/host: class Covsum, class BCM, free-function surface) compiled into a
driver that calls it the way the reference's own mains call theirs, checked against the oracle."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_cpp_dropin_driver(tmp_path, si128, golden_si128, oracle):
    X, y = si128
    ntrain = 120
    Xall = np.vstack([X[:ntrain], np.array(golden_si128["bcm"]["Xt"])])
    yall = np.concatenate([y[:ntrain], y[:3], y[:5]])
    inp, lab = tmp_path / "in.txt", tmp_path / "lab.txt"
    with open(inp, "w") as f:
        f.write("%d %d\n" % Xall.shape)
        for r in Xall:
            f.write(" ".join(repr(float(v)) for v in r) + "\n")
    with open(lab, "w") as f:
        f.write("\n".join(repr(float(v)) for v in yall) + "\n")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "cugp_amd", "host")])
    exe = str(tmp_path / "dropin_driver")
    libdir = os.path.join(ROOT, "cugp_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++14", os.path.join(ROOT, "tests", "cpp", "dropin_driver.cpp"),
                           "-o", exe, "-L" + libdir, "-lcugp_host", "-lcugp", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe, str(inp), str(lab), str(ntrain), "4"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-2000:])
    txt = out.stdout
    js = txt[txt.index("{"):]
    js = re.sub(r"\n+ PLEASE-SEE 3[^\n]*\n+", "\n", js)
    tail = js[js.rindex("}") + 1:]
    r = json.loads(js[: js.rindex("}") + 1])
    nlpp_api = float(re.search(r"NLPP = ([-+.\deE]+)", tail).group(1))        # testing_phase prints it

    o = oracle
    Xtr, ytr, Xt = Xall[:ntrain], yall[:ntrain], Xall[ntrain:]
    hp = [1.5, 1.5, 1.5]
    llo, gro = o.loglik_grad(Xtr, ytr, hp)
    assert abs(r["ll"] - llo) <= 1e-8 * abs(llo) and np.allclose(r["grad"], gro, rtol=1e-7, atol=1e-7)
    assert abs(r["api_ll"] - llo) <= 1e-8 * abs(llo) and np.allclose(r["api_grad"], gro, rtol=1e-7, atol=1e-7)
    mo, vo = o.predict(Xtr, ytr, hp, Xt)
    assert np.allclose(r["pred_mean"], mo, rtol=1e-8, atol=1e-8) and np.allclose(r["pred_var"], vo, rtol=1e-8, atol=1e-8)
    assert abs(r["nlpp"] - o.nlpp(yall[ntrain:], mo, vo)) <= 1e-8
    fin, _ = o.cg_solve(Xtr, ytr, hp)
    assert np.allclose(r["cg_final_hp"], fin, atol=5e-5) and np.allclose(r["api_cg_final_hp"], fin, atol=5e-5)
    assert abs(r["cg_final_ll"] - o.loglik(Xtr, ytr, fin)) <= 1e-5
    Ko = o.K_train(Xtr, r["cg_final_hp"])
    assert np.allclose(r["K_row5"], Ko[5], rtol=1e-12, atol=1e-14)
    assert r["param_dim"] == 2
    # rprop_solve (covkernel.cpp:337-402) through the class, compute_squared_dist, testing_phase through surface B
    rf, _ = o.rprop_solve(Xtr, ytr, hp)
    assert np.allclose(r["rprop_final_hp"], rf, atol=5e-5)
    assert abs(r["rprop_final_ll"] - o.loglik(Xtr, ytr, rf)) <= 1e-5
    So = o.sqdist(Xtr, 2.5)
    assert np.allclose(r["sqdist_row3"], So[3], rtol=1e-14, atol=0)
    assert abs(nlpp_api - o.nlpp(yall[ntrain:], mo, vo)) <= 1e-8
    b = o.bcm(Xtr, ytr, 4, hp)
    assert abs(r["bcm_ll"] - b.loglik()[0]) <= 1e-8 * abs(r["bcm_ll"])
    assert np.allclose(r["bcm_grad"], b.grad(), rtol=1e-7, atol=1e-7)
    bm, bv = b.predict(Xt)
    assert np.allclose(r["bcm_pred_mean"], bm, rtol=1e-8, atol=1e-8)
    assert np.allclose(r["bcm_pred_var"], bv, rtol=1e-8, atol=1e-8)
    bf, _ = b.cg_solve()
    assert np.allclose(r["bcm_cg_final_hp"], bf, atol=5e-5)
    # BCM passed by value (distributed_ver1.cpp:13,285): the copy drives the same device model, hyper-parameters set
    # through it are the object's, and the model survives the copy's destruction; assignment shares it too
    b2 = o.bcm(Xtr, ytr, 4, [1.25, 1.0, 0.5])
    assert r["bcm_byvalue_hp_seen"] == [1.25, 1.0, 0.5]
    assert abs(r["bcm_byvalue_ll"] - b2.loglik()[0]) <= 1e-8 * abs(r["bcm_byvalue_ll"])
    assert r["bcm_after_copy_ll"] == r["bcm_byvalue_ll"] == r["bcm_assigned_ll"]
    # surface B through the double buffers the reference's background reader fills
    assert r["api_ll_from_buffer"] == r["api_ll"]


def test_cpp_rccl_driver_one_rank(tmp_path):
    """tests/cpp/rccl_driver.cpp: the multi-process BCM driver a C++ host writes on libcugp + RCCL (one rank per GPU,
    ncclCommInitRank, device-resident K x 4 rows -> ncclAllReduce -> sum in expert order; PoE partial sums ->
    ncclAllReduce -> cugp_poe_finish), run at ONE rank on the one GPU of this box: bit-identical to the library's
    single-process sums and to the Python BCM on the same experts."""
    import cugp_amd.gp as gp
    from conftest import synth
    K, rows, d, nt = 3, 700, 10, 9
    X, y = synth(K * rows, seed=31)
    Xt = np.ascontiguousarray(X[:nt] * 0.9 + 0.05)
    hp = np.array([np.log(3.0), 0.0, np.log(0.1)])
    data = tmp_path / "data.bin"
    with open(data, "wb") as f:
        for a in (X, y, Xt, hp):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
    exe = str(tmp_path / "rccl_driver")
    libdir = os.path.join(ROOT, "cugp_amd", "lib")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-w", os.path.join(ROOT, "tests", "cpp", "rccl_driver.cpp"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + libdir, "-lcugp", "-lrccl",
                           "-Wl,-rpath," + libdir, "-o", exe])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([exe, str(tmp_path / "nccl.id"), "0", "1", str(data), str(K), str(rows), str(d), str(nt), "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-2000:])
    r = json.loads(out.stdout[out.stdout.index("{"):])
    assert r["ll"] == r["direct_ll"] and r["grad"] == r["direct_grad"]                      # bit for bit
    assert r["pred_mean"] == r["direct_pred_mean"] and r["pred_var"] == r["direct_pred_var"]
    b = gp.BCM.split(X, y, K)
    b.set_BCM_log_hyperparam(hp)
    ll, g, per = b.loglik_grad()
    m, v = b.compute_BCM_test_means_and_var(Xt)
    b.close()
    assert r["ll"] == ll and np.array_equal(r["grad"], g) and np.array_equal(r["per_expert_ll"], per)
    assert np.array_equal(r["pred_mean"], m) and np.array_equal(r["pred_var"], v)
