"""Product-of-experts ("BCM") sharded one process per GPU.

The reference shards experts over worker processes -- chunk i goes to worker i mod W
(cuda_scalingdist/cg_solver.cpp:93,166; main.cpp:101,156) -- and moves 1 (log-likelihood) or 3
(gradient) doubles per worker over TCP for every evaluation (cg_solver.cpp:72-213), plus a 3-double
hyper-parameter broadcast (:245-279).  Here every rank keeps its experts resident on its GPU, runs the
same deterministic host optimiser, and one all-reduce per evaluation carries the per-expert
[LL, g0, g1, g2] rows (RCCL over xGMI when the process group is "nccl"; gloo in the CPU tests).
Rows are summed in expert order k = 0..K-1 on every rank, so the result is bit-identical to the
single-process loop of distributed_gp/BCM.cpp:153-198 whatever the number of ranks.

The per-expert evaluator is injectable (`expert_factory`) so the sharding / reduction logic can be
exercised on CPU ranks in the tests; the default builds `cugp_amd.gp.Covsum` handles on the GPU.
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import gp as _gp


def expert_owner(k, world):
    """Reference placement: chunk k is handled by worker k mod W (cg_solver.cpp:93)."""
    return k % world


def split_rows(N, K):
    """BCM::BCM row partition (BCM.cpp:85-110): floor(N/K) rows each, remainder to the last."""
    part = N // K
    return [(k * part, part if k < K - 1 else N - part * (K - 1)) for k in range(K)]


def gather_rows_per_rank(K, world):
    """Row slots every rank contributes to the all-gather: ceil(K / W) (ranks with fewer experts leave zeros)."""
    return -(-K // world) if world > 0 else K


def gather_row_index(k, world, per):
    """Row of expert k in the gathered [W * per, 4] tensor: rank k mod W owns it as its (k // W)-th expert."""
    return expert_owner(k, world) * per + k // world


def _default_factory(n, d, device):
    return _gp.Covsum(n, d, device)


class ShardedBCM:
    """K experts over `world` ranks.  `experts` is a list of K (X_k, y_k) pairs; only the ones this
    rank owns are touched (the rest may be None).  group=None with world==1 needs no process group."""

    def __init__(self, experts, rank=0, world=1, device=0, group=None, expert_factory=None, comm_device=None):
        self.K = len(experts)
        self.rank, self.world, self.group = rank, world, group
        self.mine = [k for k in range(self.K) if expert_owner(k, world) == rank]
        factory = expert_factory or _default_factory
        self.local = {}
        self._group = None
        if expert_factory is None and len(self.mine) > 1:
            # several experts on this GPU: one library-level BCM evaluates them with shared launches
            # (csrc/bcm.cpp, group.h); self.local holds borrowed per-expert views for prediction
            data = [(np.ascontiguousarray(experts[k][0], dtype=np.float64),
                     np.ascontiguousarray(experts[k][1], dtype=np.float64)) for k in self.mine]
            self._group = _gp.BCM([X.shape[0] for X, _ in data], data[0][0].shape[1], device)
            for i, (X, y) in enumerate(data):
                self._group.set_expert_data(i, X, y)
                self.local[self.mine[i]] = self._group.expert(i)
        else:
            for k in self.mine:
                X, y = experts[k]
                X = np.ascontiguousarray(X, dtype=np.float64)
                y = np.ascontiguousarray(y, dtype=np.float64)
                e = factory(X.shape[0], X.shape[1], device)
                e.set_data(X, y)
                self.local[k] = e
        self.hp = np.zeros(3)
        # host-clock seconds spent in this rank's evaluations (enqueue -> rows in place) and in the exchange
        # (staging copy, all-reduce, copy back), summed since reset_timers(): what a multi-GPU run is diagnosed from
        self.t_device = self.t_collective = 0.0
        if comm_device is None:
            comm_device = torch.device("cuda", device) if (world > 1 and dist.get_backend(group) == "nccl") \
                else torch.device("cpu")
        self.comm_device = comm_device
        self._rows = torch.zeros((self.K, 4), dtype=torch.float64, device=comm_device)
        self._send = torch.zeros((self.K, 4), dtype=torch.float64, device=comm_device)
        if comm_device.type == "cuda":
            torch.cuda.current_stream(comm_device).synchronize()      # the zeros are there before the library writes rows
        # RCCL path: the per-expert rows go from the evaluation's result buffer straight into this device tensor
        # (no fetch / numpy / H2D on the critical path); needs the library-level BCM of this rank's experts
        self._on_device = (comm_device.type == "cuda" and expert_factory is None and len(self.mine) > 0)
        if self._on_device and self._group is None:
            X, y = (np.ascontiguousarray(a, dtype=np.float64) for a in experts[self.mine[0]])
            for e in self.local.values():
                e.close()
            self._group = _gp.BCM([X.shape[0]], X.shape[1], device)
            self._group.set_expert_data(0, X, y)
            self.local = {self.mine[0]: self._group.expert(0)}
        # Lean exchange (round 6): the library writes this rank's rows into a compact [per, 4] device tensor, ONE
        # all-gather moves them (no zero rows, no staging copy: all_reduce is in place, so the reduce form needs a fresh
        # copy of the send buffer every evaluation), the result is copied into PINNED memory without blocking and one
        # stream synchronise ends the evaluation.  Rank r owns experts r, r + W, ...: `per` = ceil(K / W) row slots per
        # rank, unused ones stay zero.  CUGP_BCM_EXCHANGE=allreduce selects the round-5 form (A/B runs).
        # (The CPU / gloo path takes the same all-gather with host tensors, so the world-size-2 CPU test covers its layout.)
        self._per = gather_rows_per_rank(self.K, world)
        # Exchange forms (CUGP_BCM_EXCHANGE overrides; all give the same bits):
        #   library    the whole exchange inside libcugp (csrc/comm.cpp): ncclAllGather on the evaluation's own stream
        #              directly behind its last kernel, copy into pinned memory behind that, ONE host wait -- no host
        #              round trip between evaluation and collective, no torch ops.  Default where the rows are on the
        #              device and the process group is RCCL (or there is one rank).
        #   allgather  the same gather through torch.distributed (gloo on the CPU; device tensors moved by gloo in tests)
        #   allreduce  rounds 1-5: K x 4 zero-padded rows, staging copy + all_reduce + blocking copy back
        backend = dist.get_backend(group) if (world > 1 or (dist.is_available() and dist.is_initialized())) else None
        form = os.environ.get("CUGP_BCM_EXCHANGE", "")
        if not form:
            form = "library" if (comm_device.type == "cuda" and expert_factory is None and (world == 1 or backend == "nccl")) else "allgather"
        self.exchange_form = form
        self._comm = None
        if form == "library":
            # rank 0's id reaches the other ranks through the process group that already exists
            idt = torch.zeros(_gp.Comm.ID_BYTES, dtype=torch.uint8)
            rehearse = world == 1 and backend == "nccl"           # one rank, but go through a real communicator
            if world > 1 or rehearse:
                if rank == 0:
                    idt = torch.frombuffer(bytearray(_gp.Comm.unique_id()), dtype=torch.uint8).clone()
                if world > 1:
                    idt = idt.to(comm_device if backend == "nccl" else "cpu")
                    dist.broadcast(idt, src=0, group=group)
                uid = bytes(idt.cpu().numpy().tobytes())
            else:
                uid = None
            try:
                self._comm = _gp.Comm(uid, rank, world, device)
            except Exception as exc:                      # (librccl not loadable, communicator refused, ...)
                self._comm, self._comm_error = None, exc
            if world > 1:
                # every rank takes the same form: if the library's communicator failed anywhere, all ranks fall back
                # to the all-gather through torch.distributed (the same RCCL, driven from Python) and say so
                ok = torch.tensor([1 if self._comm is not None else 0], dtype=torch.int32,
                                  device=comm_device if backend == "nccl" else "cpu")
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
                if int(ok.item()) == 0:
                    if self._comm is not None:
                        self._comm.close()
                        self._comm = None
                    form = self.exchange_form = "allgather"
                    if rank == 0:
                        print("cugp_amd.bcm: the library's RCCL communicator could not be created on every rank (%s); "
                              "using the all-gather through torch.distributed" % getattr(self, "_comm_error", "another rank"),
                              flush=True)
            elif self._comm is None:
                raise self._comm_error
        self._lean = form == "allgather"
        if self._lean:
            self._mine_dev = torch.zeros((self._per, 4), dtype=torch.float64, device=comm_device)
            self._all_dev = torch.zeros((world * self._per, 4), dtype=torch.float64, device=comm_device)
            self._slots = list(range(len(self.mine)))               # expert mine[i] -> row i of _mine_dev
            if comm_device.type == "cuda":
                self._all_host = torch.zeros((world * self._per, 4), dtype=torch.float64).pin_memory()
                torch.cuda.current_stream(comm_device).synchronize()
            else:
                self._all_host = self._all_dev

    # BCM::set_BCM_log_hyperparam (BCM.cpp:123-130): every expert gets the same vector
    def set_loghyper(self, hp):
        self.hp = np.array(hp, dtype=np.float64)
        if self._group is not None:
            self._group.set_BCM_log_hyperparam(self.hp)
            return
        for e in self.local.values():
            e.set_loghyperparam(self.hp)

    def _allreduce(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def _allgather(self, out, mine):
        if self.world > 1:
            dist.all_gather_into_tensor(out, mine, group=self.group)
        else:
            out.copy_(mine)
        return out

    def reset_timers(self):
        self.t_device = self.t_collective = 0.0

    def loglik_grad(self):
        """-> (sum_k LL_k, sum_k grad_k, per-expert LL[K]); one collective of K x 4 doubles."""
        t0 = time.perf_counter()
        if self._comm is not None:
            # (device and collective time are not separable on the host clock here: they are one stream sequence;
            #  t_collective keeps what the host spends beyond the call)
            g = self._comm.loglik_grad_allgather(self._group, self._per)
            t1 = time.perf_counter()
            out = np.stack([g[gather_row_index(k, self.world, self._per)] for k in range(self.K)]) if self.K else np.zeros((0, 4))
            self.t_device += t1 - t0
            self.t_collective += time.perf_counter() - t1
            return self._ordered_sum(out)
        if self._lean:
            if self._on_device:
                self._group.loglik_grad_rows_device(self._mine_dev.data_ptr(), self._slots)   # returns with the rows in place
            else:
                rows = self._local_rows()
                if len(self.mine):
                    self._mine_dev[:len(self.mine)].copy_(torch.from_numpy(rows[self.mine]))
            t1 = time.perf_counter()
            self._allgather(self._all_dev, self._mine_dev)
            if self._all_host is not self._all_dev:
                self._all_host.copy_(self._all_dev, non_blocking=True)
                torch.cuda.current_stream(self.comm_device).synchronize()
            g = self._all_host.numpy()
            out = np.stack([g[gather_row_index(k, self.world, self._per)] for k in range(self.K)]) if self.K else np.zeros((0, 4))
            self.t_device += t1 - t0
            self.t_collective += time.perf_counter() - t1
            return self._ordered_sum(out)
        if self._on_device:
            # _send: zero everywhere except this rank's rows, which every evaluation overwrites (the other ranks' rows
            # must be exact zeros in the sum); the collective works on a copy, so nothing has to be cleared or waited
            # for on the host between evaluations
            self._group.loglik_grad_rows_device(self._send.data_ptr(), self.mine)   # returns with the rows in place
            t1 = time.perf_counter()
            self._rows.copy_(self._send)
            out = self._allreduce(self._rows).cpu().numpy()
            self.t_device += t1 - t0
            self.t_collective += time.perf_counter() - t1
            return self._ordered_sum(out)
        rows = self._local_rows()
        t1 = time.perf_counter()
        self._rows.copy_(torch.from_numpy(rows))
        out = self._allreduce(self._rows).cpu().numpy()
        self.t_device += t1 - t0
        self.t_collective += time.perf_counter() - t1
        return self._ordered_sum(out)

    def _local_rows(self):
        """[K, 4] host rows: this rank's experts' (LL, gradient), zeros elsewhere."""
        rows = np.zeros((self.K, 4))
        if self._group is not None:
            rows[self.mine] = self._group.loglik_grad_rows()
        else:
            for k in self.mine:                   # all local experts in flight before the first fetch
                self.local[k].enqueue(True)
            for k in self.mine:
                ll, g = self.local[k].fetch()
                rows[k, 0] = ll
                rows[k, 1:] = g
        return rows

    def _ordered_sum(self, out):
        ll, g = 0.0, np.zeros(3)
        for k in range(self.K):                   # expert order, as BCM.cpp:161-197
            ll = ll + out[k, 0]
            g = out[k, 1:].copy() if k == 0 else g + out[k, 1:]
        return float(ll), g, out[:, 0].copy()

    def predict(self, Xt):
        """Product of experts (BCM.cpp:45-83): all-reduce of per-expert precision and precision*mean."""
        Xt = np.ascontiguousarray(Xt, dtype=np.float64)
        nt = Xt.shape[0]
        buf = np.zeros((self.K, 2, nt))
        for k in self.mine:
            m, v = self.local[k].compute_test_means_and_variances(None, None, Xt)
            buf[k, 0] = 1.0 / v
            buf[k, 1] = (1.0 / v) * m
        t = torch.from_numpy(buf).to(self.comm_device)
        out = self._allreduce(t).cpu().numpy()
        sp, spm = np.zeros(nt), np.zeros(nt)
        for k in range(self.K):
            sp += out[k, 0]
            spm += out[k, 1]
        return _gp.poe_finish(sp, spm)

    def objective(self, theta):
        self.set_loghyper(theta)
        ll, g, _ = self.loglik_grad()
        return -1.0 * ll, g

    def cg_solve(self, budget=100):
        """cg_solve(BCM) (distributed_ver1.cpp:13-232) -- the library's host loop on the all-reduced
        objective; every rank runs it on identical numbers, so no hyper-parameter broadcast."""
        theta, trace = _gp.cg_minimize(self.objective, self.hp, budget)
        self.set_loghyper(theta)
        return trace

    def close(self):
        for e in self.local.values():
            if hasattr(e, "close"):
                e.close()
        self.local = {}
        if self._group is not None:
            self._group.close()
            self._group = None
        if getattr(self, "_comm", None) is not None:
            self._comm.close()
            self._comm = None
