"""N > 1 path on CPU: two gloo ranks run cugp_amd.bcm.ShardedBCM (the real sharding, all-reduce and
host CG code) with the per-expert evaluator swapped for the oracle, and must reproduce the
single-process product-of-experts numbers of the reference bit for bit."""
import os
import socket
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class OracleExpert:
    """Stand-in for gp.Covsum with the same enqueue/fetch/predict surface, backed by the CPU oracle."""

    def __init__(self, n, d, device):
        from oracle.oracle_py import Oracle
        self.o = Oracle()
        self.hp = np.zeros(3)

    def set_data(self, X, y):
        self.X, self.y = X, y

    def set_loghyperparam(self, hp):
        self.hp = np.array(hp, dtype=np.float64)

    def enqueue(self, want_grad=True):
        pass

    def fetch(self):
        return self.o.loglik(self.X, self.y, self.hp), self.o.grad(self.X, self.y, self.hp)

    def compute_test_means_and_variances(self, X, y, Xt):
        return self.o.predict(self.X, self.y, self.hp, Xt)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cugp_amd.bcm import ShardedBCM, split_rows
    d = np.load(os.path.join(ROOT, "tests", "golden", "data_si128.npz"))
    X, y = d["X"], d["y"]
    K = 4
    experts = [(X[o:o + n], y[o:o + n]) for o, n in split_rows(128, K)]
    b = ShardedBCM(experts, rank=rank, world=world, expert_factory=OracleExpert)
    assert b.mine == [k for k in range(K) if k % world == rank]
    b.set_loghyper([1.5, 1.5, 1.5])
    ll, g, per = b.loglik_grad()
    Xt = np.vstack([X[:3], X[:5] * 0.7 - 0.1])
    m, v = b.predict(Xt)
    tr = b.cg_solve(30)
    # an UNEVEN shard count (5 experts on 2 ranks: rank 0 owns 3, rank 1 owns 2 and leaves a zero slot in the
    # all-gather) against the same experts summed in this process without any collective
    K5 = 5
    ex5 = [(X[o:o + n], y[o:o + n]) for o, n in split_rows(128, K5)]
    b5 = ShardedBCM(ex5, rank=rank, world=world, expert_factory=OracleExpert)
    b5.set_loghyper([1.2, 0.7, -0.3])
    ll5, g5, per5 = b5.loglik_grad()
    s5 = ShardedBCM(ex5, rank=0, world=1, expert_factory=OracleExpert)
    s5.set_loghyper([1.2, 0.7, -0.3])
    sl5, sg5, sper5 = s5.loglik_grad()
    assert ll5 == sl5 and np.array_equal(g5, sg5) and np.array_equal(per5, sper5), (rank, ll5, sl5)
    os.environ["CUGP_BCM_EXCHANGE"] = "allreduce"      # the round-5 exchange gives the same bits
    b5r = ShardedBCM(ex5, rank=rank, world=world, expert_factory=OracleExpert)
    b5r.set_loghyper([1.2, 0.7, -0.3])
    ll5r, g5r, per5r = b5r.loglik_grad()
    os.environ.pop("CUGP_BCM_EXCHANGE")
    assert ll5r == sl5 and np.array_equal(g5r, sg5) and np.array_equal(per5r, sper5)
    q.put((rank, ll, g, per, m, v, b.hp.copy(), tr))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bcm_matches_reference_golden():
    import json
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_si128.json")))["bcm"]
    r0, r1 = res
    # both ranks hold identical all-reduced numbers
    assert r0[1] == r1[1] and np.array_equal(r0[2], r1[2]) and np.array_equal(r0[6], r1[6])
    assert np.array_equal(r0[7], r1[7])
    # and they equal the reference's single-process BCM (same per-expert values, same summation order)
    assert abs(r0[1] - gold["ll"]) <= 1e-12 * abs(gold["ll"])
    assert np.allclose(r0[2], gold["grad"], rtol=1e-13, atol=1e-13)
    assert np.allclose(r0[3], gold["ll_per_expert_6dp"], atol=5e-7)
    assert np.allclose(r0[4], gold["pred_mean"], rtol=1e-13, atol=1e-13)
    assert np.allclose(r0[5], gold["pred_var"], rtol=1e-13, atol=1e-13)
    # 30-evaluation CG run == the first 30 probes of the reference's cg_solve(BCM) trace
    probes = np.array([p[1:] for p in gold["cg"]["please_see"] if p[0] in (1, 2)])
    tr = r0[7]
    n = min(tr.shape[0] - 1, probes.shape[0])
    assert n >= 20 and np.allclose(tr[1:1 + n, :3], probes[:n], atol=6e-7)
