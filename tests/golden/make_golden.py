#!/usr/bin/env python3
"""Generate tests/golden/* from the reference itself.  Run HERE (the container
with /root/reference); the GPU box only ever sees the committed outputs.

    python tests/golden/make_golden.py            # small cases  (~2 min)
    python tests/golden/make_golden.py --big 4096 # adds sine_4096 LL+grad (~15 min, 1 core)
    python tests/golden/make_golden.py --big 8192 # adds siproper_8192 LL+grad (~2.5 h, 1 core)

Inputs  : the reference's data files (chunked_dataset/*.txt) -> data_*.npz
Outputs : expected values computed by the reference's own C++ compiled where it
          lies (oracle/Makefile -> oracle/_ref/libref_*.so) -> golden_*.json
Also    : the numeric trace of the reference's committed run log
          cuda_bettersinglenode_ver2/REF -> ref_log_si128.json

Round 4 (prediction at size, optimiser trajectories): --job list4 names them; pred8192_* take ~2.3 h each, si6000_poe
~3.4 h, the others minutes to an hour.

Round 2 (configs 3-5 and a dense metric-size case; CPU-hours, so one job per
process, see run_jobs.sh):
    python tests/golden/make_golden.py --job data        # data_si24000.npz, data_siproper_10000.npz
    python tests/golden/make_golden.py --job NAME        # -> golden_r2/NAME.json  (NAME: see JOBS)

Only data (inputs, expected outputs) is written; no reference source text.
"""
import argparse
import json
import os
import re
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("CUGP_REFERENCE", "/root/reference")
DS = os.path.join(REF, "chunked_dataset")

HP_DEFAULT = [0.5, 0.5, 0.5]                      # cpp_serial_gp/serial_gp.cpp:49
HP_BCM = [1.5, 1.5, 1.5]                          # distributed_gp/distributed_ver1.cpp:274
HP_DENSE = [3.762111, -1.152105, -0.384461]       # cuda_src/main.cpp:190-193


def load_txt(prefix, rows=None):
    X = np.loadtxt(os.path.join(DS, prefix + "_chunk0.txt"), skiprows=1, max_rows=rows)
    y = np.loadtxt(os.path.join(DS, prefix + "_label0.txt"), max_rows=rows)
    return np.ascontiguousarray(X), np.ascontiguousarray(y)


def parse_please_see(path):
    """-> list of [kind, hp0, hp1, hp2] from a cg_solve stdout capture."""
    out = []
    pat = re.compile(r"PLEASE-SEE\s+(\d)\s*:\s*([-\d.eE+naninf]+),\s*([-\d.eE+naninf]+),\s*([-\d.eE+naninf]+)")
    with open(path, errors="replace") as f:
        for line in f:
            m = pat.search(line)
            if m:
                out.append([int(m.group(1))] + [float(m.group(i)) for i in (2, 3, 4)])
    return out


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)
    print("wrote", name)


def ll_grad(r, X, y, hp):
    t0 = time.time()
    ll = r.loglik(X, y, hp)
    t1 = time.time()
    g = r.grad(X, y, hp)
    t2 = time.time()
    return {"hp": list(hp), "n": int(X.shape[0]), "ll": ll, "grad": g.tolist(),
            "t_ll_s": round(t1 - t0, 4), "t_grad_s": round(t2 - t1, 4)}


def small(r):
    tmp = tempfile.mkdtemp()
    # ---- data fixtures -------------------------------------------------
    Xs = np.vstack([np.loadtxt(os.path.join(DS, f"si32_chunk{i}.txt"), skiprows=1) for i in range(4)])
    ys = np.concatenate([np.loadtxt(os.path.join(DS, f"si32_label{i}.txt")) for i in range(4)])
    np.savez_compressed(os.path.join(HERE, "data_si128.npz"), X=Xs, y=ys)
    Xq, yq = load_txt("sine_dataset_4096_10", 4096 + 64)      # 4096 train rows + 64 spare (test points)
    np.savez_compressed(os.path.join(HERE, "data_sine_4160.npz"), X=Xq, y=yq)

    # ---- si128: LL/grad/K/L/inverse/predict/NLPP/CG/BCM ----------------
    g = {"dataset": "chunked_dataset/si32_chunk{0..3}.txt concatenated (= REF's input_128)", "cases": []}
    for hp in (HP_BCM, HP_DEFAULT, HP_DENSE):
        c = ll_grad(r, Xs, ys, hp)
        K = r.K_train(Xs, hp)
        L = r.cholesky(K)
        Ki = r.K_inverse(K)
        q, ld = r.chol_and_det(K, ys)
        Xt = Xs[:3] * 0.9 + 0.05
        m, v = r.predict(Xs, ys, hp, Xt)
        c.update({"K_row5": K[5].tolist(), "K_diag": np.diag(K).tolist(), "L_row100": L[100].tolist(),
                  "L_diag": np.diag(L).tolist(), "Kinv_row7": Ki[7].tolist(), "Kinv_trace": float(np.trace(Ki)),
                  "quad": q, "logdet": ld, "Xt": Xt.tolist(), "pred_mean": m.tolist(), "pred_var": v.tolist(),
                  "nlpp": r.nlpp(ys[:3], m, v)})
        g["cases"].append(c)
    # serial cg_solve from hp=1.5 (the REF run) and hp=0.5 (serial_gp.cpp default)
    g["cg"] = []
    for hp in (HP_BCM, HP_DEFAULT):
        log = os.path.join(tmp, "cg.log")
        final = r.cg_solve(Xs, ys, hp, log)
        g["cg"].append({"hp0": list(hp), "final_hp": final.tolist(), "final_ll": r.loglik(Xs, ys, final),
                        "please_see": parse_please_see(log)})
    log = os.path.join(tmp, "rprop.log")
    final = r.rprop_solve(Xs, ys, HP_BCM, log)
    g["rprop"] = {"hp0": HP_BCM, "final_hp": final.tolist(), "final_ll": r.loglik(Xs, ys, final)}
    # BCM, 4 experts x 32 rows (distributed_ver1.cpp:279-285)
    b = r.bcm(Xs, ys, 4, HP_BCM)
    log = os.path.join(tmp, "bcm_ll.log")
    ll = b.loglik(log)
    per = [float(m.group(1)) for m in re.finditer(r"LL of Expert \d+: ([-\d.]+)", open(log).read())]
    Xt = np.vstack([Xs[:3], Xs[:5] * 0.7 - 0.1])
    m, v = b.predict(Xt)
    bc = {"K": 4, "hp": HP_BCM, "ll": ll, "ll_per_expert_6dp": per, "grad": b.grad().tolist(),
          "Xt": Xt.tolist(), "pred_mean": m.tolist(), "pred_var": v.tolist(),
          "nlpp": b.nlpp(np.concatenate([ys[:3], ys[:5]]), m, v)}
    # uneven split: 3 experts over 128 rows -> 42,42,44 (BCM.cpp:92-108)
    b3 = r.bcm(Xs, ys, 3, HP_DENSE)
    m3, v3 = b3.predict(Xt)
    bc["uneven"] = {"K": 3, "hp": HP_DENSE, "ll": b3.loglik(), "grad": b3.grad().tolist(),
                    "pred_mean": m3.tolist(), "pred_var": v3.tolist()}
    log = os.path.join(tmp, "bcm_cg.log")
    b2 = r.bcm(Xs, ys, 4, HP_BCM)
    final = b2.cg_solve(log)
    bc["cg"] = {"hp0": HP_BCM, "final_hp": final.tolist(), "please_see": parse_please_see(log)}
    g["bcm"] = bc
    dump("golden_si128.json", g)

    # ---- sine (D=10): 256 / 1024 / 2048 rows ----------------------------
    s = {"dataset": "chunked_dataset/sine_dataset_4096_10_chunk0.txt, first n rows "
                    "(the sine_dataset_{256,1024,2048}_10 files hold the same leading rows)", "cases": []}
    X, y = Xq[:256], yq[:256]
    for hp in (HP_DEFAULT, HP_DENSE):
        c = ll_grad(r, X, y, hp)
        Xt = Xq[4096:4096 + 8]
        m, v = r.predict(X, y, hp, Xt)
        c.update({"test_rows": [4096, 4104], "pred_mean": m.tolist(), "pred_var": v.tolist(),
                  "nlpp": r.nlpp(yq[4096:4104], m, v)})
        s["cases"].append(c)
    log = os.path.join(tmp, "cg256.log")
    final = r.cg_solve(X, y, HP_DENSE, log)
    s["cg256"] = {"hp0": HP_DENSE, "final_hp": final.tolist(), "final_ll": r.loglik(X, y, final),
                  "please_see": parse_please_see(log)}
    b = r.bcm(Xq[:1024], yq[:1024], 4, HP_DENSE)
    m, v = b.predict(Xq[4096:4104])
    s["bcm1024x4"] = {"K": 4, "hp": HP_DENSE, "ll": b.loglik(), "grad": b.grad().tolist(),
                      "test_rows": [4096, 4104], "pred_mean": m.tolist(), "pred_var": v.tolist()}
    for n in (1024, 2048):
        s["cases"].append(ll_grad(r, Xq[:n], yq[:n], HP_DENSE))
        print("  n =", n, s["cases"][-1]["ll"], s["cases"][-1]["t_ll_s"], s["cases"][-1]["t_grad_s"])
    dump("golden_sine.json", s)

    # ---- the reference's committed run log (REF) ------------------------
    refp = os.path.join(REF, "cuda_bettersinglenode_ver2", "REF")
    txt = open(refp, errors="replace").read()
    lls = [float(x) for x in re.findall(r"The value of loglikelihood = ([-\d.]+)", txt)]
    grads = [[float(a), float(b_), float(c_)] for a, b_, c_ in
             re.findall(r"Final gradients of log hyperparams are ([-\d.]+), ([-\d.]+), ([-\d.]+)", txt)]
    ps = parse_please_see(refp)
    dump("ref_log_si128.json", {"source": "cuda_bettersinglenode_ver2/REF (reference's committed stdout, 6 d.p.)",
                                "hp0": HP_BCM, "loglik_values": lls, "gradients": grads, "please_see": ps})


def big(r, n):
    name = "golden_big_%d.json" % n          # one file per size: the two runs may overlap in time
    path = os.path.join(HERE, name)
    cur = {"cases": {}}
    if n == 4096:
        d = np.load(os.path.join(HERE, "data_sine_4160.npz"))
        X, y, hp, key = d["X"][:4096], d["y"][:4096], HP_DENSE, "sine_4096"
    elif n == 8192:
        X, y = load_txt("siproper_9192_10")                 # 8192 train + 1000 test, ver2/main.cpp:178-181
        np.savez_compressed(os.path.join(HERE, "data_siproper_9192.npz"), X=X, y=y)
        X, y, hp, key = X[:8192], y[:8192], HP_DEFAULT, "siproper_8192"
    else:
        raise SystemExit("--big takes 4096 or 8192")
    t0 = time.time()
    ll = r.loglik(X, y, hp)
    t1 = time.time()
    cur["cases"][key] = {"hp": hp, "n": n, "ll": ll, "t_ll_s": round(t1 - t0, 2)}
    dump(name, cur)
    g = r.grad(X, y, hp)
    t2 = time.time()
    cur = json.load(open(path))
    cur["cases"][key].update({"grad": g.tolist(), "t_grad_s": round(t2 - t1, 2)})
    dump(name, cur)


# ---------------------------------------------------------------------------
# round 2: configs 3-5 on the reference's own data, dense K at the metric size
# ---------------------------------------------------------------------------
SC = os.path.join(REF, "scaling_dataset")
HP_TWO = [2.0, 2.0, 2.0]                          # cuda_scalingdist/main.cpp:298-301
R2 = os.path.join(HERE, "golden_r2")
HP_TAIL = [0.882908, 0.098703, -2.971479]         # cuda_bettersinglenode_ver2/REF:3167,3183 (end of the CG run)
HP_ILL = [3.762111, 0.098703, -2.971479]          # HP_DENSE's length scale, the tail's amplitude and noise


def make_data():
    """data_si24000.npz (configs 4+5 share rows) and data_siproper_10000.npz (config 3)."""
    X = np.loadtxt(os.path.join(SC, "si24000_all_input.txt"), delimiter=",")
    y = np.loadtxt(os.path.join(SC, "si24000_all_label.txt"))
    # the chunk files hold the same rows: 4 x 6000 (chunked_dataset) and 16 x 1500 (scaling_dataset)
    for k in range(4):
        Xk = np.loadtxt(os.path.join(DS, "si6000_chunk%d.txt" % k), skiprows=1)
        yk = np.loadtxt(os.path.join(DS, "si6000_label%d.txt" % k))
        assert np.array_equal(Xk, X[6000 * k:6000 * (k + 1)]) and np.array_equal(yk, y[6000 * k:6000 * (k + 1)]), k
    for k in range(16):
        Xk = np.loadtxt(os.path.join(SC, "si24000_16sharded_chunk%d.txt" % k), skiprows=1)
        yk = np.loadtxt(os.path.join(SC, "si24000_16sharded_label%d.txt" % k))
        assert np.array_equal(Xk, X[1500 * k:1500 * (k + 1)]) and np.array_equal(yk, y[1500 * k:1500 * (k + 1)]), k
    np.savez_compressed(os.path.join(HERE, "data_si24000.npz"), X=X, y=y)
    X, y = load_txt("siproper_10000_10")
    assert X.shape == (10000, 10)
    np.savez_compressed(os.path.join(HERE, "data_siproper_10000.npz"), X=X, y=y)
    print("wrote data_si24000.npz, data_siproper_10000.npz")


def _rows(which):
    if which == "d8192":
        d = np.load(os.path.join(HERE, "data_siproper_9192.npz"))
        return d["X"][:8192], d["y"][:8192]
    if which == "s10000":
        d = np.load(os.path.join(HERE, "data_siproper_10000.npz"))
        return d["X"], d["y"]
    d = np.load(os.path.join(HERE, "data_si24000.npz"))
    m = re.match(r"si6000_(\d)$", which)
    if m:
        k = int(m.group(1))
        return d["X"][6000 * k:6000 * (k + 1)], d["y"][6000 * k:6000 * (k + 1)]
    return d["X"], d["y"]


def job(r, name):
    """One reference computation -> golden_r2/<name>.json."""
    os.makedirs(R2, exist_ok=True)
    t0 = time.time()
    if name == "si24000_bcm16":
        # config 5: 16 x 1500 rows, in-memory BCM (BCM.cpp:85-110 partitions rows exactly like the chunk files)
        X, y = _rows("si24000")
        out = {"K": 16, "rows": [0, 24000], "cases": []}
        Xt = np.vstack([X[:4] * 0.9 + 0.05, X[12000:12004] * 0.8 - 0.1])
        yt = np.concatenate([y[:4], y[12000:12004]])
        for hp in (HP_DENSE, HP_TWO):
            b = r.bcm(X, y, 16, hp)
            log = os.path.join(tempfile.mkdtemp(), "ll.log")
            ll = b.loglik(log)
            per = [float(m.group(1)) for m in re.finditer(r"LL of Expert \d+: ([-\d.]+)", open(log).read())]
            g = b.grad()
            m, v = b.predict(Xt)
            out["cases"].append({"hp": hp, "ll": ll, "ll_per_expert_6dp": per, "grad": g.tolist(), "Xt": Xt.tolist(),
                                 "yt": yt.tolist(), "pred_mean": m.tolist(), "pred_var": v.tolist(),
                                 "nlpp": b.nlpp(yt, m, v)})
    elif name in ("si24000_bcm16_tail", "si24000_bcm16_ill"):
        # round 3: config 5 at the hyper-parameters the reference's CG ends at (REF:3167-3183), and ill-conditioned
        X, y = _rows("si24000")
        HP_T = HP_TAIL if name.endswith("tail") else HP_ILL
        b = r.bcm(X, y, 16, HP_T)
        log = os.path.join(tempfile.mkdtemp(), "ll.log")
        ll = b.loglik(log)
        per = [float(m.group(1)) for m in re.finditer(r"LL of Expert \d+: ([-\d.]+)", open(log).read())]
        out = {"K": 16, "rows": [0, 24000], "hp": HP_T, "ll": ll, "ll_per_expert_6dp": per,
               "grad": b.grad().tolist()}
    elif name == "cg_sine1024":
        # round 3: a cg_solve trajectory above 256 rows (covkernel.cpp:405-647)
        d = np.load(os.path.join(HERE, "data_sine_4160.npz"))
        X, y = d["X"][:1024], d["y"][:1024]
        log = os.path.join(tempfile.mkdtemp(), "cg.log")
        final = r.cg_solve(X, y, HP_DENSE, log)
        out = {"rows": "sine_1024", "n": 1024, "hp0": HP_DENSE, "final_hp": final.tolist(),
               "final_ll": r.loglik(X, y, final), "please_see": parse_please_see(log)}
    elif name in ("pred4096", "pred8192_dense", "pred8192_ill"):
        # round 4: predictive mean / variance / NLPP of a single GP at size (covkernel.cpp:105-116,277-323,649-659)
        if name == "pred4096":
            d = np.load(os.path.join(HERE, "data_sine_4160.npz"))
            n, t0r, t1r, hp, rows = 4096, 4096, 4160, HP_DENSE, "sine_4160"
        else:
            d = np.load(os.path.join(HERE, "data_siproper_9192.npz"))
            n, t0r, t1r, rows = 8192, 8192, 8192 + 150, "siproper_9192"
            hp = HP_DENSE if name.endswith("dense") else HP_ILL
        X, y = d["X"][:n], d["y"][:n]
        Xt, yt = d["X"][t0r:t1r], d["y"][t0r:t1r]
        m, v = r.predict(X, y, hp, Xt)
        out = {"rows": rows, "n": n, "hp": hp, "test_rows": [t0r, t1r], "pred_mean": m.tolist(),
               "pred_var": v.tolist(), "nlpp": r.nlpp(yt, m, v)}
    elif name == "si6000_poe":
        # round 4: config 4's product-of-experts prediction, 4 x 6000 rows (BCM.cpp:45-83)
        X, y = _rows("si24000")
        Xt = np.vstack([X[::600] * 0.9 + 0.05, X[300::1200] * 0.8 - 0.1])
        yt = np.concatenate([y[::600], y[300::1200]])
        b = r.bcm(X, y, 4, HP_DENSE)
        m, v = b.predict(Xt)
        out = {"K": 4, "rows": [0, 24000], "hp": HP_DENSE, "Xt": Xt.tolist(), "yt": yt.tolist(),
               "pred_mean": m.tolist(), "pred_var": v.tolist(), "nlpp": b.nlpp(yt, m, v)}
    elif name in ("cg_sine2048", "rprop_sine1024"):
        # round 4: optimiser trajectories at larger sizes (covkernel.cpp:337-402, 405-647)
        d = np.load(os.path.join(HERE, "data_sine_4160.npz"))
        n = 2048 if name == "cg_sine2048" else 1024
        X, y = d["X"][:n], d["y"][:n]
        log = os.path.join(tempfile.mkdtemp(), "opt.log")
        final = (r.cg_solve if name.startswith("cg") else r.rprop_solve)(X, y, HP_DENSE, log)
        out = {"rows": "sine_%d" % n, "n": n, "hp0": HP_DENSE, "final_hp": final.tolist(),
               "final_ll": r.loglik(X, y, final), "please_see": parse_please_see(log)}
    elif name == "bcm16_cg_8000":
        # round 4: cg_solve(BCM) (distributed_ver1.cpp:13-232) on 16 experts x 500 rows (rows 0..7999 of si24000)
        X, y = _rows("si24000")
        X, y = X[:8000], y[:8000]
        b = r.bcm(X, y, 16, HP_DENSE)
        log = os.path.join(tempfile.mkdtemp(), "bcmcg.log")
        final = b.cg_solve(log)
        b2 = r.bcm(X, y, 16, final.tolist())
        out = {"K": 16, "rows": [0, 8000], "hp0": HP_DENSE, "final_hp": final.tolist(), "final_ll": b2.loglik(),
               "please_see": parse_please_see(log)}
    elif name == "bcm5_6007":
        # round 4: an uneven in-memory BCM at a middle size: 6007 rows in 5 experts -> 1201 x 4 + 1203 (BCM.cpp:85-110:
        # the remainder goes to the last expert), likelihood, gradient, product-of-experts prediction, NLPP
        X, y = _rows("si24000")
        X, y = X[9000:9000 + 6007], y[9000:9000 + 6007]
        Xt = np.vstack([X[::250] * 0.9 + 0.05, X[100::500] * 0.8 - 0.1])
        yt = np.concatenate([y[::250], y[100::500]])
        out = {"K": 5, "rows": [9000, 15007], "cases": []}
        for hp in (HP_DENSE, HP_TWO):
            b = r.bcm(X, y, 5, hp)
            log = os.path.join(tempfile.mkdtemp(), "ll.log")
            ll = b.loglik(log)
            per = [float(m.group(1)) for m in re.finditer(r"LL of Expert \d+: ([-\d.]+)", open(log).read())]
            g = b.grad()
            m, v = b.predict(Xt)
            out["cases"].append({"hp": hp, "ll": ll, "ll_per_expert_6dp": per, "grad": g.tolist(), "Xt": Xt.tolist(),
                                 "yt": yt.tolist(), "pred_mean": m.tolist(), "pred_var": v.tolist(), "nlpp": b.nlpp(yt, m, v)})
    elif re.match(r"(tail|ill)(\d+)_(ll|grad)$", name):
        # round 3: the ill-conditioned regime.  "tail" = REF's end point on sine rows; "ill" = the dense
        # length scale of HP_DENSE with the tail's amplitude and noise (cond(K) ~ n*sf2/sn2)
        m = re.match(r"(tail|ill)(\d+)_(ll|grad)$", name)
        n = int(m.group(2))
        d = np.load(os.path.join(HERE, "data_sine_4160.npz"))
        X, y = d["X"][:n], d["y"][:n]
        hp = HP_TAIL if m.group(1) == "tail" else HP_ILL
        out = {"rows": "sine_%d" % n, "n": n, "hp": hp}
        if m.group(3) == "ll":
            out["ll"] = r.loglik(X, y, hp)
            if n <= 4096:
                K = r.K_train(X, hp)
                w = np.linalg.eigvalsh(K)
                out["cond_K"] = float(w[-1] / w[0])
        else:
            out["grad"] = r.grad(X, y, hp).tolist()
    else:
        m = re.match(r"(d8192|s10000|si6000_\d)_(ll|grad)(_two|_ill)?$", name)
        if not m:
            raise SystemExit("unknown job " + name)
        X, y = _rows(m.group(1))
        hp = HP_TWO if m.group(3) == "_two" else (HP_ILL if m.group(3) == "_ill" else HP_DENSE)
        out = {"rows": m.group(1), "n": int(X.shape[0]), "hp": hp}
        if m.group(2) == "ll":
            out["ll"] = r.loglik(X, y, hp)
        else:
            out["grad"] = r.grad(X, y, hp).tolist()
    out["t_s"] = round(time.time() - t0, 1)
    with open(os.path.join(R2, name + ".json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote golden_r2/%s.json in %.0f s" % (name, out["t_s"]))


JOBS = (["s10000_grad", "d8192_grad", "s10000_ll"] + ["si6000_%d_grad" % k for k in range(4)] + ["d8192_ll"]
        + ["si6000_%d_ll" % k for k in range(4)] + ["si24000_bcm16"])      # longest first
JOBS_R3 = ["tail4096_grad", "ill4096_grad", "si24000_bcm16_tail", "tail4096_ll", "ill4096_ll", "cg_sine1024",
           "tail2048_grad", "ill2048_grad", "tail2048_ll", "ill2048_ll",
           "d8192_grad_ill", "d8192_ll_ill", "si24000_bcm16_ill"]           # round 3 (written into golden_r2/ too)
JOBS_R4 = ["si6000_poe", "pred8192_dense", "pred8192_ill", "pred4096",     # round 4: prediction at size
           "cg_sine2048", "rprop_sine1024", "bcm16_cg_8000", "bcm5_6007"]                 # ... and optimiser trajectories

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", type=int, default=0)
    ap.add_argument("--job", default="")
    a = ap.parse_args()
    from oracle.oracle_py import Reference, build
    if a.job == "list":
        print("\n".join(JOBS))
        raise SystemExit(0)
    if a.job == "list3":
        print("\n".join(JOBS_R3))
        raise SystemExit(0)
    if a.job == "list4":
        print("\n".join(JOBS_R4))
        raise SystemExit(0)
    if a.job == "data":
        make_data()
        raise SystemExit(0)
    build(ref=True)
    ref = Reference()
    if a.job:
        job(ref, a.job)
    elif a.big:
        big(ref, a.big)
    else:
        small(ref)
