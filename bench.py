#!/usr/bin/env python3
"""bench.py -- GP log-lik+grad evaluations/s at N=8192, D=10 (fp64) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one evaluation of the hot path at a fresh hyper-parameter vector: SE-kernel build,
blocked Cholesky, triangular inverse, K^-1, log-likelihood and its 3 gradients (the pair the
reference's cg_solve calls at every probe, covkernel.cpp:500-501).  Inputs are synthetic
(X ~ U(-10,10)^10, y = sin(x0) + 0.1 N(0,1)) and RESIDENT IN HBM before the timed region.
With N ranks every rank owns one N=8192 expert (BCM sharding: experts are independent, weak scaling)
and the per-evaluation exchange is one all-reduce of K x 4 doubles over RCCL.

Rank 0 prints ONE JSON line; besides the driver's contract it carries
  roofline     the Cholesky trailing update (fp64 MFMA SYRK), HIP-event timed per launch in the timed region
  cpu_baseline the serial CPU restatement of cpp_serial_gp (oracle/, 1 thread) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one hardware queue per stream for up to 16 experts on a GPU (runtime default: 4); read at HIP initialisation,
# so it is set before torch touches the device (libcugp.so sets the same default when it is loaded first)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

N_METRIC, D_METRIC = 8192, 10
HP0 = np.array([np.log(3.0), 0.0, np.log(0.1)])      # non-degenerate point: K is dense, cond ~ 1e3
MFMA_F64_PEAK_TFLOPS = 78.6                          # MI355X dense fp64 matrix peak (spec)


def synth(n, d, seed):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-10.0, 10.0, (n, d))
    y = np.sin(X[:, 0]) + 0.1 * rng.standard_normal(n)
    return np.ascontiguousarray(X), np.ascontiguousarray(y)


def cpu_baseline(n_sample, n_full, d):
    """Oracle (reported baseline only): LL + gradient, 1 thread, first n_sample rows of the workload."""
    from oracle.oracle_py import Oracle
    o = Oracle()
    X, y = synth(n_full, d, 15618)
    X, y = np.ascontiguousarray(X[:n_sample]), np.ascontiguousarray(y[:n_sample])
    t0 = time.perf_counter()
    ll = o.loglik(X, y, HP0)
    t1 = time.perf_counter()
    g = o.grad(X, y, HP0)
    t2 = time.perf_counter()
    sec = t2 - t0
    scale = (n_full / n_sample) ** 3                  # flop-proportional; flatters the CPU (its measured exponent is >3)
    return {
        "value": 1.0 / (sec * scale), "unit": "evals/s", "cores": 1, "kind": "port",
        "sample": "LL+grad on the first %d rows of the same synthetic workload: %.2f s (LL %.2f s, grad %.2f s); "
                  "scaled to N=%d by (N/n)^3" % (n_sample, sec, t1 - t0, t2 - t1, n_full),
        "ll_sample": ll, "grad_sample": [float(v) for v in g],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", dest="n", type=int, default=N_METRIC, help="rows per expert (metric config: 8192)")
    ap.add_argument("--dims", dest="d", type=int, default=D_METRIC)
    ap.add_argument("--experts-per-gpu", type=int, default=1)
    ap.add_argument("--experts-total", type=int, default=0,
                    help="strong-scaling BCM workload: this many experts in total, expert k on rank k mod N "
                         "(e.g. --experts-total 16 --rows 1500 = the si24000 16-shard shape); value = BCM objective "
                         "evaluations/s (all experts + all-reduce per evaluation)")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="rows for the CPU baseline (0 = skip)")
    ap.add_argument("--overlap", type=int, default=1, help="0: build the inverse after the factorisation on one stream "
                    "(every kernel has the chip to itself: the per-kernel roofline of the whole run is the isolated one "
                    "and no extra pass is made)")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from cugp_amd.bcm import ShardedBCM

    strong = args.experts_total > 0
    K = args.experts_total if strong else world * args.experts_per_gpu
    experts = [None] * K
    for k in range(K):
        if k % world == rank:
            experts[k] = synth(args.n, args.d, 15618 + k)
    bcm = ShardedBCM(experts, rank=rank, world=world, device=local_rank)
    timed_launches = len(bcm.local) == 1              # per-launch HIP events: the single-expert (metric) workload;
    for e in bcm.local.values():                      # several experts per GPU share launches and are not timed singly
        if timed_launches:
            e.set_profiling(2)
        if not args.overlap:
            e.set_overlap(False)

    def step(i):
        bcm.set_loghyper(HP0 + 1e-3 * ((i % 7) - 3))          # fresh hyper-parameters: nothing is cached
        return bcm.loglik_grad()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    if timed_launches:
        first_e = next(iter(bcm.local.values()))
        first_e.kernel_stats(reset=True)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ll, g, _ = step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    first = next(iter(bcm.local.values()))
    ks = first.kernel_stats() if timed_launches else {"launches": 0}
    ph = first.phase_ms() if timed_launches else {"potrf": float("nan")}
    npad = -(-args.n // 128) * 128

    # Outside the timed region: the same kernel with the chip to itself.  In the timed region the inverse blocks
    # run beside the factorisation on other streams, so a trailing-update launch shares the CUs and its duration
    # is not a statement about the kernel alone; eight more evaluations with the overlap off give that number.
    iso = iso_ph = None
    if rank == 0 and len(bcm.local) == 1 and args.overlap:
        first.set_overlap(False)
        first.kernel_stats(reset=True)
        for i in range(8):                        # level-2 profiling times every 8th launch, rotating
            first.set_loghyperparam(HP0 + 1e-3 * ((i % 7) - 3))
            first.loglik_grad()
        iso = first.kernel_stats()
        iso_ph = first.phase_ms()
        first.set_overlap(True)

    if rank == 0:
        evals = args.steps * (1 if strong else K)
        out = {
            "metric": ("BCM log-lik+grad evals/sec (%d experts x N=%d, D=%d)" % (K, args.n, args.d)) if strong
            else "GP log-lik+grad evals/sec (N=%d, D=%d)" % (args.n, args.d),
            "value": evals / dt, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("bcm_%dx%d_D%d" % (K, args.n, args.d)) if strong
                       else "gp_loglik_grad_N%d_D%d" % (args.n, args.d), "experts": K,
                       "experts_per_gpu": (K + world - 1) // world if strong else args.experts_per_gpu,
                       "sharding": "bcm-experts-per-gpu", "overlap": bool(args.overlap) and len(bcm.local) == 1,
                       "hp": HP0.tolist()},
            "cholesky_gflops": (npad ** 3 / 3.0) / ((iso_ph or ph)["potrf"] * 1e-3) / 1e9,   # factorisation alone (overlap off)
            "eval_tflops_n3": (float(args.n) ** 3) * (K if strong else world * args.experts_per_gpu) / world
                              / (dt / args.steps) / 1e12,
            "phase_ms_last": {k: round(v, 4) for k, v in ph.items()},
            "ll_last": ll, "grad_last": [float(v) for v in g],
        }
        traffic = None      # HBM bytes per launch from the PMC passes committed under profiles/ (tools/pmc.sh,
        try:                # tools/pmc_summary.py: 2 x FETCH_SIZE + WRITE_SIZE, per launch); not collectable in-process
            with open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")) as f:
                if args.n == N_METRIC:
                    traffic = json.load(f)["kernels"]["k_syrk_step"]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        if ks["launches"] > 0:
            ach = ks["flop"] / (ks["sum_ms"] * 1e-3) / 1e12
            out["roofline"] = {
                "kernel": "k_syrk_step (Cholesky trailing update, fp64 MFMA 16x16x4, K=128 per launch)",
                "bound": "mfma", "achieved": ach, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / MFMA_F64_PEAK_TFLOPS, "traffic": traffic,
                "launches": int(ks["launches"]), "avg_launch_us": 1e3 * ks["sum_ms"] / ks["launches"],
                "algorithmic_flop_per_launch": ks["flop"] / ks["launches"],
            }
            if iso is not None:
                out["roofline"]["note"] = ("achieved/frac: timed region, where a launch shares the CUs with the inverse "
                                           "blocks on the other streams; isolated_*: same kernel, overlap off")
            if iso and iso["launches"] > 0:
                ia = iso["flop"] / (iso["sum_ms"] * 1e-3) / 1e12
                out["roofline"].update({"isolated_achieved": ia, "isolated_frac": ia / MFMA_F64_PEAK_TFLOPS,
                                        "isolated_avg_launch_us": 1e3 * iso["sum_ms"] / iso["launches"]})
            out["roofline"]["whole_evaluation_frac"] = out["eval_tflops_n3"] / MFMA_F64_PEAK_TFLOPS
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(min(args.cpu_sample, args.n), args.n, args.d)
        print(json.dumps(out), flush=True)

    bcm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
