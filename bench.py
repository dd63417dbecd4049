#!/usr/bin/env python3
"""bench.py -- GP log-lik+grad evaluations/s at N=8192, D=10 (fp64) on MI355X.

    python bench.py [--steps K] [--warmup W]                    (one GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W   (N > 1: one rank per GPU; under torchrun
                                                                  --gpus defaults to WORLD_SIZE)

One "step" = one evaluation of the hot path at a fresh hyper-parameter vector: SE-kernel build,
blocked Cholesky, triangular inverse, K^-1, log-likelihood and its 3 gradients (the pair the
reference's cg_solve calls at every probe, covkernel.cpp:500-501).  Inputs are synthetic
(X ~ U(-10,10)^10, y = sin(x0) + 0.1 N(0,1)) and RESIDENT IN HBM before the timed region.
With N ranks every rank owns one N=8192 expert (BCM sharding: experts are independent, weak scaling)
and the per-evaluation exchange is one all-reduce of K x 4 doubles over RCCL.

The run is made of passes (--passes, default all three, in this order):
  timed      W warm-up + K evaluations on the DEFAULT path (no per-launch events, no instrumentation): `value`
  profiled   the same K evaluations with every launch of every MFMA kernel (and the covariance build) timed by its own
             workgroups (cugp_set_profiling 5: first workgroup's start / last workgroup's end on the chip-wide 100 MHz
             clock, stamped into a device buffer; no events, nothing added to the streams): the `roofline*` records --
             the timed pass's schedule, the inverse blocks beside the factorisation, so a launch shares the CUs with
             kernels on the other streams
  isolated   8 evaluations with the overlap off (every kernel has the chip to itself): `isolated_*`, cholesky_gflops
tools/make_profiles.sh runs rocprofv3 --kernel-trace --stats once per pass (--passes timed | profiled | isolated), so
the average duration of a kernel in the CSV named in roofline.dominant_by reproduces roofline.achieved.

Rank 0 prints ONE JSON line; besides the driver's contract it carries
  roofline        the MFMA kernel with the largest share of kernel time in the profiled pass (HIP events around its
                  launches on the stream they run on; algorithmic flop / duration against the fp64 MFMA peak)
  roofline_kernels  the same record for every timed kernel, named as rocprofv3 names them
  roofline_trailing_update  k_syrk_step + k_syrk_wide together: the N^3/3 flop of the factorisation's trailing update
  roofline_kbuild the SE-kernel build: bytes written / HIP-event time of its launch against the HBM peak
  predict         1000 test points against the N=8192 model (covkernel.cpp:277-323), ms per call
  bcm_si24000_16shard, bcm_si6000_4chunk   BASELINE configs 5 and 4 as strong-scaling sub-runs in the same process
                  group (16 x 1500 and 4 x 6000 rows, expert k on rank k mod N): ms per BCM evaluation, evals/s
  cpu_baseline    the serial CPU restatement of cpp_serial_gp (oracle/, 1 thread) on a bounded sample, next to
                  what the reference's own code, compiled unmodified, needed for the full-size evaluation
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
HBM_PEAK_GBS = 8000.0                                # MI355X HBM3E (MI355X_MICROARCH.md)


def pmc_summary_for(build_id):
    """The committed counter summary (profiles/<tag>_pmc_summary.json: rocprofv3 --pmc passes of this command,
    tools/make_profiles.sh) that was measured on THIS library: its `build_id` is the hash of the sources the library was
    built from (cugp_build_id).  -> (tag, relative path, kernels dict) or (None, None, reason)."""
    import glob
    seen = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), reverse=True):
        try:
            with open(f) as fh:
                d = json.load(fh)
        except Exception:
            continue
        seen.append("%s: %s" % (os.path.basename(f), d.get("build_id")))
        if d.get("build_id") == build_id:
            return os.path.basename(f)[:-len("_pmc_summary.json")], os.path.relpath(f, ROOT), d["kernels"]
    return None, None, ("no committed counter summary was measured on the loaded library (build id %s; summaries: %s): "
                        "run tools/make_profiles.sh <tag> on the GPU box and commit profiles/<tag>_pmc_summary.json"
                        % (build_id, "; ".join(seen) or "none"))

# kernels timed by the library (cugp_get_kernel_stats_kind): name as rocprofv3 prints it, what it is, and one launch in
# how many is timed at profiling level 2 (the isolated pass; the profiled pass runs level 3: every launch)
KINDS = {0: ("k_syrk_step", "Cholesky near-window trailing update + next diagonal block, K=128 per launch", 16),
         1: ("k_syrk_wide", "Cholesky far trailing update, K=128*panel per launch", 1),
         2: ("k_trtri_border<4>", "bordering steps of L^-1, 128x128 output tiles", 4),
         3: ("k_trtri_border<2>", "bordering steps of L^-1, 64x64 output tiles", 4),
         4: ("k_lauum<4>", "shares of K^-1 = L^-T L^-1, 128x128 output tiles", 4),
         5: ("k_lauum<2>", "shares of K^-1, 64x64 output tiles", 4),
         6: ("k_trtri_level<4>", "doubling inside a block of inverse rows, 128x128 output tiles", 16),
         7: ("k_trtri_level<2>", "doubling inside a block of inverse rows, 64x64 output tiles", 16),
         8: ("k_trtri_block", "a hand-over block's own inverse in one launch: diagonal tiles + all doubling levels "
                              "(latency-bound: <= 64 workgroups, stage barriers)", 4)}


def synth(n, d, seed):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-10.0, 10.0, (n, d))
    y = np.sin(X[:, 0]) + 0.1 * rng.standard_normal(n)
    return np.ascontiguousarray(X), np.ascontiguousarray(y)


def host_cpu():
    """(model string of the host CPU, logical cores of the box)"""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return model, os.cpu_count()


def cpu_baseline(n_sample, n_full, d):
    """Oracle (reported baseline only): LL + gradient, 1 thread PINNED to one core, first
    n_sample rows of the workload; the host CPU and its core count are reported next to the number."""
    from oracle.oracle_py import Oracle
    o = Oracle()
    X, y = synth(n_full, d, 15618)
    X, y = np.ascontiguousarray(X[:n_sample]), np.ascontiguousarray(y[:n_sample])
    model, ncores = host_cpu()
    allowed, pinned = None, None
    try:                                              # one core of the set this process may run on (the last: away from
        allowed = os.sched_getaffinity(0)             # the cores the runtime's helper threads were started on)
        pinned = max(allowed)
        os.sched_setaffinity(0, {pinned})
    except (AttributeError, OSError):
        pinned = None
    try:
        t0 = time.perf_counter()
        ll = o.loglik(X, y, HP0)
        t1 = time.perf_counter()
        g = o.grad(X, y, HP0)
        t2 = time.perf_counter()
    finally:
        if allowed is not None and pinned is not None:
            os.sched_setaffinity(0, allowed)
    sec = t2 - t0
    scale = (n_full / n_sample) ** 3                  # flop-proportional; flatters the CPU (its measured exponent is >3)
    ref_s = None                                      # the reference compiled unmodified, full size, build container
    try:                                              # (tests/golden/make_golden.py --big 8192: data, not code)
        with open(os.path.join(ROOT, "tests", "golden", "golden_big_8192.json")) as f:
            c = json.load(f)["cases"]["siproper_8192"]
        if n_full == N_METRIC:
            ref_s = c["t_ll_s"] + c["t_grad_s"]
    except Exception:
        ref_s = None
    return {
        "value": 1.0 / (sec * scale), "unit": "evals/s", "cores": 1, "kind": "port",
        "host_cpu": model, "host_cores_total": ncores, "pinned_core": pinned,
        "cores_allowed": len(allowed) if allowed is not None else None,
        "sample": "LL+grad on the first %d rows of the same synthetic workload: %.2f s (LL %.2f s, grad %.2f s); "
                  "scaled to N=%d by (N/n)^3" % (n_sample, sec, t1 - t0, t2 - t1, n_full),
        "ll_sample": ll, "grad_sample": [float(v) for v in g],
        "reference_compiled_s": ref_s,
        "reference_compiled_note": "cpp_serial_gp compiled from the reference's sources (oracle/Makefile), LL + gradient "
                                   "on siproper_9192 rows 0..8191, one core of the build container" if ref_s else None,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="ranks = GPUs (default: WORLD_SIZE under torchrun, else 1)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", dest="n", type=int, default=N_METRIC, help="rows per expert (metric config: 8192)")
    ap.add_argument("--dims", dest="d", type=int, default=D_METRIC)
    ap.add_argument("--experts-per-gpu", type=int, default=1)
    ap.add_argument("--experts-total", type=int, default=0,
                    help="strong-scaling BCM workload: this many experts in total, expert k on rank k mod N "
                         "(e.g. --experts-total 16 --rows 1500 = the si24000 16-shard shape); value = BCM objective "
                         "evaluations/s (all experts + all-reduce per evaluation)")
    ap.add_argument("--cpu-sample", type=int, default=2560, help="rows for the CPU baseline (0 = skip)")
    ap.add_argument("--sub-steps", type=int, default=10, help="timed evaluations of each BCM sub-run (configs 4, 5); 0 = skip")
    ap.add_argument("--overlap", type=int, default=1, help="0: build the inverse after the factorisation on one stream "
                    "(every kernel has the chip to itself: the per-kernel roofline of the whole run is the isolated one "
                    "and no extra pass is made)")
    ap.add_argument("--passes", default="timed,profiled,isolated", help="comma list of timed, profiled, isolated (see the "
                    "module docstring); the contract line needs `timed`")
    ap.add_argument("--prof-level", type=int, default=5, help="profiling level of the profiled / isolated passes: 5 = every "
                    "timed launch's own workgroups stamp their first start / last end (s_memrealtime) into a device buffer -- "
                    "nothing is added to the streams, the schedule is the timed pass's; 4 = every MFMA launch carries its own "
                    "start / stop events (hipExtLaunchKernelGGL: +3 %% per evaluation, the start event still sits in front of "
                    "the dispatch gap); 3 = an event pair recorded around every such launch (rounds 3-4: ~6 us of dispatch gap "
                    "inside the bracket, ~5 us of device time per pair)")
    ap.add_argument("--tune", default="", help="comma list key=value of launch-shape keys (cugp_amd/csrc/kernels.h TUNE_*) set as "
                    "process defaults before any handle exists -- A/B runs only; the contract line is measured without it")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--rehearse-rccl", action="store_true", help="rehearsal on ONE GPU of the code path the ranks of a "
                    "multi-GPU run take: a 1-rank RCCL process group, the expert's row written into the device tensor and "
                    "all-reduced there (cugp_bcm_loglik_grad_rows_device)")
    args = ap.parse_args()

    # stdout carries ONE line, the JSON record.  Whatever the libraries underneath print while they initialise (RCCL
    # writes a version banner to stdout when its first communicator is created) goes to stderr: file descriptor 1 points
    # at stderr until the record is printed.
    sys.stdout.flush()
    fd_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus <= 0:
        args.gpus = world
    if world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d (WORLD_SIZE is %d)"
                         % (args.gpus, args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    collective = world > 1 or args.rehearse_rccl
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)   # RCCL saw all N ranks

    from cugp_amd.bcm import ShardedBCM
    if args.tune:
        from cugp_amd import capi
        for kv in args.tune.split(","):
            k, v = kv.split("=")
            capi.check(capi.lib().cugp_set_tuning(int(k), int(v)))

    strong = args.experts_total > 0
    K = args.experts_total if strong else world * args.experts_per_gpu
    experts = [None] * K
    for k in range(K):
        if k % world == rank:
            experts[k] = synth(args.n, args.d, 15618 + k)
    bcm = ShardedBCM(experts, rank=rank, world=world, device=local_rank,
                     comm_device=torch.device("cuda", local_rank) if args.rehearse_rccl else None)
    if args.rehearse_rccl and world == 1:
        assert bcm._on_device
        bcm._allreduce = lambda t: (dist.all_reduce(t, op=dist.ReduceOp.SUM), t)[1]    # the collective, at one rank
        bcm._allgather = lambda o, m: (dist.all_gather_into_tensor(o, m), o)[1]
    if not bcm.local and strong and K < world:
        pass                                          # more ranks than experts: this rank only takes part in the collectives
    passes = [p for p in args.passes.split(",") if p]
    can_profile = K == world                          # per-launch HIP events: one expert on EVERY rank (the metric workload;
    for e in bcm.local.values():                      # the same on all ranks: the passes hold barriers); several experts per GPU share launches
        e.set_profiling(0)
        if not args.overlap:
            e.set_overlap(False)

    def step(i):
        bcm.set_loghyper(HP0 + 1e-3 * ((i % 7) - 3))          # fresh hyper-parameters: nothing is cached
        return bcm.loglik_grad()

    def fence():
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(nwarm, nsteps):
        """nwarm untimed + nsteps timed evaluations; -> (max-over-ranks seconds, last ll, last gradient)"""
        for i in range(nwarm):
            step(i)
        fence()
        bcm.reset_timers()
        t0_ = time.perf_counter()
        r_ = (float("nan"), np.full(3, np.nan), None)
        for i in range(nsteps):
            r_ = step(nwarm + i)
        fence()
        dt_ = time.perf_counter() - t0_
        tm_ = torch.tensor([dt_], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        if collective:
            dist.all_reduce(tm_, op=dist.ReduceOp.MAX)
        return float(tm_.item()), r_[0], r_[1]

    first = next(iter(bcm.local.values()), None)      # None: a rank that owns no expert
    empty = {"launches": 0, "sum_ms": 0.0, "flop": 0.0, "disp_ms": 0.0}
    nanph = {"potrf": float("nan"), "kbuild": float("nan")}

    # ---- pass 1, timed: the default path, nothing instrumented -> `value`
    dt, ll, g = float("nan"), float("nan"), np.full(3, np.nan)
    exchange = None
    if "timed" in passes:
        dt, ll, g = timed_steps(args.warmup, args.steps)
        # this rank's split of an evaluation on the host clock: its experts (enqueue -> rows in device memory), the
        # exchange (collective + read-back into pinned memory + synchronise), and what is left (hyper-parameter
        # update, Python) -- the record a rehearsal of one rank of an 8-GPU run is read from
        exchange = {"form": bcm.exchange_form, "backend": args.backend if collective else None,
                    "ranks_in_group": world if collective else 0,
                    "device_ms": 1e3 * bcm.t_device / args.steps, "collective_ms": 1e3 * bcm.t_collective / args.steps,
                    "host_between_evaluations_ms": 1e3 * (dt - bcm.t_device - bcm.t_collective) / args.steps}

    # ---- pass 2, profiled: the same evaluations with HIP events around the MFMA launches -> roofline*
    timed_launches = can_profile and "profiled" in passes
    kst = {kd: dict(empty) for kd in KINDS}
    kb_st = None
    ph = dict(nanph)
    dt_prof = None
    if timed_launches:
        first.set_profiling(args.prof_level)          # EVERY launch of the MFMA kernels timed
        step(0)                                       # event pools, first profiled enqueue
        for kd in KINDS:
            first.kernel_stats(reset=True, kind=kd)
        dt_prof, ll_p, g_p = timed_steps(0, args.steps)
        kst = {kd: first.kernel_stats(kind=kd) for kd in KINDS}
        kb_st = first.kernel_stats(kind=10) if args.prof_level >= 5 else None
        ph = first.phase_ms()
        if "timed" not in passes:
            dt, ll, g = dt_prof, ll_p, g_p
    npad = -(-args.n // 128) * 128

    # ---- pass 3, isolated: the same kernels with the chip to themselves.  In the other passes the inverse blocks
    # run beside the factorisation on other streams, so a launch shares the CUs and its duration is not a statement
    # about the kernel alone; eight more evaluations with the overlap off give that number.
    iso = iso_ph = None
    if rank == 0 and can_profile and args.overlap and "isolated" in passes:
        first.set_overlap(False)
        first.set_profiling(1)                    # phase events only: the factorisation alone -> cholesky_gflops
        for i in range(4):                        # (events around every launch would add ~5 us to each of its 126)
            first.set_loghyperparam(HP0 + 1e-3 * ((i % 7) - 3))
            first.loglik_grad()
        iso_ph = first.phase_ms()
        first.set_profiling(args.prof_level)      # ... and every MFMA launch with the chip to itself
        first.set_loghyperparam(HP0)
        first.loglik_grad()
        for kd in KINDS:
            first.kernel_stats(reset=True, kind=kd)
        for i in range(8):
            first.set_loghyperparam(HP0 + 1e-3 * ((i % 7) - 3))
            first.loglik_grad()
        iso = {kd: first.kernel_stats(kind=kd) for kd in KINDS}
        first.set_overlap(True)
    if first is not None and can_profile:
        first.set_profiling(0)

    # prediction (covkernel.cpp:277-323): 1000 test points against the resident model, factor and inverse valid
    predict = None
    if rank == 0 and world == 1 and len(bcm.local) == 1 and not strong:
        Xt = synth(1000, args.d, 99)[0]
        first.set_profiling(0)
        first.set_loghyperparam(HP0)
        first.loglik_grad()
        first.compute_test_means_and_variances(None, None, Xt)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for _ in range(5):
            first.compute_test_means_and_variances(None, None, Xt)
        torch.cuda.synchronize()
        pms = 1e3 * (time.perf_counter() - tp) / 5
        predict = {"what": "predictive mean + variance of 1000 test points against the N=%d model (host arrays in, "
                           "host arrays out; factor and inverse resident)" % args.n, "ntest": 1000, "ms": pms,
                   "points_per_s": 1000.0 / (pms * 1e-3)}
        # its MFMA kernel: W = Ks L^-T, test tile x row tile ti sums the k tiles <= ti (the diagonal k tile of the
        # triangular L^-1 counted half): ntest_tiles * nt^2 / 2 * 2 * 128^3 flop -- HALF of the square product's
        first.set_profiling(args.prof_level)
        first.kernel_stats(reset=True, kind=9)
        for _ in range(5):
            first.compute_test_means_and_variances(None, None, Xt)
        pst = first.kernel_stats(kind=9)
        first.set_profiling(0)
        if pst["launches"] > 0:
            pach = pst["flop"] / (pst["sum_ms"] * 1e-3) / 1e12
            predict["roofline"] = {"kernel": "k_predict_gemm (W = Ks L^-T, 64x64 output tiles in pairs of equal total K; fp64 MFMA 16x16x4)",
                                   "bound": "mfma", "achieved": pach, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": pach / MFMA_F64_PEAK_TFLOPS, "avg_launch_us": 1e3 * pst["sum_ms"] / pst["launches"],
                                   "algorithmic_flop_per_launch": pst["flop"] / pst["launches"],
                                   "launches_timed": int(pst["launches"]),
                                   "note": "triangular operand: 8 test tiles x 64 row tiles x (ti + 1/2) k tiles = 6.87e10 flop "
                                           "at N=8192, 1000 test points -- half of 2*1024*8192^2"}

    # BASELINE configs 5 and 4 as strong-scaling sub-runs in the same process group: the si24000 16-shard and the
    # si6000 4-chunk shapes (synthetic rows), expert k on rank k mod N (cuda_scalingdist/cg_solver.cpp:93,166), one
    # all-reduce of K x 4 doubles per evaluation -- so that a 1/2/4/8-GPU run yields the north_star's scaling curve
    bcm.close()
    subs = {}
    if args.sub_steps > 0 and not strong and args.n == N_METRIC and args.experts_per_gpu == 1:
        for name, (Ks, rows) in (("bcm_si24000_16shard", (16, 1500)), ("bcm_si6000_4chunk", (4, 6000))):
            ex = [synth(rows, args.d, 24000 + 100 * Ks + k) if k % world == rank else None for k in range(Ks)]
            sb = ShardedBCM(ex, rank=rank, world=world, device=local_rank,
                            comm_device=torch.device("cuda", local_rank) if args.rehearse_rccl else None)
            if args.rehearse_rccl and world == 1:
                sb._allreduce = lambda t: (dist.all_reduce(t, op=dist.ReduceOp.SUM), t)[1]
                sb._allgather = lambda o, m: (dist.all_gather_into_tensor(o, m), o)[1]
            for i in range(2):
                sb.set_loghyper(HP0 + 1e-3 * i)
                sb.loglik_grad()
            fence()
            sb.reset_timers()
            ts = time.perf_counter()
            for i in range(args.sub_steps):
                sb.set_loghyper(HP0 + 1e-3 * ((i % 7) - 3))
                sll, sg, _ = sb.loglik_grad()
            fence()
            sdt = time.perf_counter() - ts
            # per rank: host-clock ms per evaluation in its own experts (enqueue -> rows in place) and in the exchange
            # (staging copy + all-reduce + copy back, which also absorbs waiting for the slowest rank)
            per = torch.tensor([1e3 * sb.t_device / args.sub_steps, 1e3 * sb.t_collective / args.sub_steps],
                               dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            if collective:
                allp = [torch.zeros_like(per) for _ in range(world)]
                dist.all_gather(allp, per)
            else:
                allp = [per]
            allp = [[float(v) for v in t.cpu()] for t in allp]
            tm = torch.tensor([sdt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            if collective:
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            sdt = float(tm.item())
            subs[name] = {"experts": Ks, "rows_per_expert": rows, "experts_per_gpu": -(-Ks // world), "n_gpus": world,
                          "steps": args.sub_steps, "ms_per_eval": 1e3 * sdt / args.sub_steps,
                          "evals_per_s": args.sub_steps / sdt, "scaling": "strong", "ll_last": sll,
                          "device_ms": [round(v[0], 4) for v in allp], "collective_ms": [round(v[1], 4) for v in allp],
                          "eval_tflops_n3": Ks * float(rows) ** 3 / (sdt / args.sub_steps) / 1e12}
            sb.close()

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
                       "sharding": "bcm-experts-per-gpu", "overlap": bool(args.overlap) and can_profile,
                       "hp": HP0.tolist(), "passes": passes, "tune": args.tune or None,
                       "value_from": "timed pass: default path, no per-launch events" if "timed" in passes
                       else "profiled pass (no timed pass was run)"},
            "cholesky_gflops": (npad ** 3 / 3.0) / ((iso_ph or ph)["potrf"] * 1e-3) / 1e9,   # factorisation alone (overlap off)
            "eval_tflops_n3": (float(args.n) ** 3) * (K if strong else world * args.experts_per_gpu) / world
                              / (dt / args.steps) / 1e12,
            "phase_ms_last": {k: round(v, 4) for k, v in ph.items()},
            "ms_per_step_profiled": (1e3 * dt_prof / args.steps) if dt_prof else None,
            "ll_last": ll, "grad_last": [float(v) for v in g],
        }
        if exchange:
            out["exchange"] = exchange
        if predict:
            out["predict"] = predict
        out.update(subs)
        # HBM bytes per launch from the rocprofv3 --pmc passes of this same command, committed under profiles/
        # (tools/make_profiles.sh, tools/pmc_summary.py: 2 x FETCH_SIZE + WRITE_SIZE per launch; counters cannot be
        # read in-process, so the figure is a committed measurement).  It is only reported when it was measured on the
        # library that is loaded now: the summary carries the hash of the sources it was built from (cugp_build_id);
        # otherwise traffic is null and traffic_source says why.
        from cugp_amd import capi as _capi
        build_id = (_capi.lib().cugp_build_id() or b"unknown").decode()
        out["library_build_id"] = build_id
        ROUND, PMC_SUMMARY, pmc = pmc_summary_for(build_id)
        pmc_reason = None
        if ROUND is None:
            pmc_reason, pmc = pmc, {}
        elif args.n != N_METRIC:
            pmc_reason, pmc = "counter summary is of the N=%d workload" % N_METRIC, {}

        def dur(st):
            # level 5: from the end of the launch in front of it on its stream (where rocprofv3 puts an in-order
            # dispatch's begin) to the launch's last workgroup; other levels: the event pair's bracket
            return st["disp_ms"] if st.get("disp_ms", 0.0) > 0.0 else st["sum_ms"]

        def roof(kd, st, iso_st):
            name, what, every = KINDS[kd]
            ach = st["flop"] / (dur(st) * 1e-3) / 1e12
            r = {"kernel": "%s (%s; fp64 MFMA 16x16x4)" % (name, what), "bound": "mfma", "achieved": ach,
                 "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_F64_PEAK_TFLOPS,
                 "traffic": pmc.get(name, {}).get("hbm_bytes_per_launch"),
                 "traffic_source": PMC_SUMMARY if name in pmc else (pmc_reason or "kernel not in %s" % PMC_SUMMARY),
                 "mfma_busy_frac_pmc": pmc.get(name, {}).get("mfma_busy_frac"),
                 "launches_timed": int(st["launches"]), "timed_one_launch_in": 1,
                 "avg_launch_us": 1e3 * dur(st) / st["launches"],
                 "avg_first_to_last_workgroup_us": 1e3 * st["sum_ms"] / st["launches"],
                 "algorithmic_flop_per_launch": st["flop"] / st["launches"],
                 "est_ms_per_eval": dur(st) / args.steps}
            if iso_st and iso_st["launches"] > 0:
                ia = iso_st["flop"] / (dur(iso_st) * 1e-3) / 1e12
                r.update({"isolated_achieved": ia, "isolated_frac": ia / MFMA_F64_PEAK_TFLOPS,
                          "isolated_avg_launch_us": 1e3 * dur(iso_st) / iso_st["launches"]})
            return r

        recs = {KINDS[kd][0]: roof(kd, kst[kd], iso[kd] if iso else None) for kd in KINDS if kst[kd]["launches"] > 0}
        if recs:
            tot = sum(r["est_ms_per_eval"] for r in recs.values())
            for r in recs.values():
                r["share_of_timed_kernel_time"] = r["est_ms_per_eval"] / tot
            dom = max(recs, key=lambda k: recs[k]["est_ms_per_eval"])
            out["roofline"] = dict(recs[dom])
            out["roofline"]["dominant_by"] = ("largest share of the MFMA kernels' time in the profiled pass "
                                              "(est_ms_per_eval); rocprofv3 --kernel-trace --stats of that pass alone "
                                              "(bench.py --passes profiled): profiles/<tag>_bench_profiled_n8192_kernel_stats.csv, "
                                              "of the timed pass: profiles/<tag>_bench_timed_n8192_kernel_stats.csv, of the "
                                              "isolated pass: profiles/<tag>_bench_isolated_n8192_kernel_stats.csv; <tag> = "
                                              + (ROUND or "(none measured on this library yet)"))
            out["roofline"]["note"] = ("achieved/frac: profiled pass = the timed pass's schedule with every launch timed "
                                       "(profiling level %d: 5 = by its own workgroups' first start / last end, nothing added "
                                       "to the streams), where a launch shares the CUs with kernels on the other streams; "
                                       "isolated_*: same kernel, overlap off" % args.prof_level)
            out["roofline"]["whole_evaluation_frac"] = out["eval_tflops_n3"] / MFMA_F64_PEAK_TFLOPS
            out["roofline_kernels"] = recs
            ks, kw = kst[0], kst[1]
            if ks["launches"] > 0 and kw["launches"] > 0:
                # the two kernels of the factorisation's trailing update together: their flop over the sum of their
                # durations (profiled pass: every launch of both is timed)
                fl = kw["flop"] + ks["flop"]
                ms = dur(kw) + dur(ks)
                tu = {"what": "k_syrk_wide + k_syrk_step together (N^3/3 flop of the factorisation)",
                      "achieved": fl / (ms * 1e-3) / 1e12, "unit": "TFLOP/s", "peak": MFMA_F64_PEAK_TFLOPS,
                      "wide_share_of_flop": kw["flop"] / fl}
                tu["frac"] = tu["achieved"] / MFMA_F64_PEAK_TFLOPS
                if iso and iso[0]["launches"] > 0 and iso[1]["launches"] > 0:
                    ia = (iso[1]["flop"] + iso[0]["flop"]) / ((dur(iso[1]) + dur(iso[0])) * 1e-3) / 1e12
                    tu["isolated_achieved"], tu["isolated_frac"] = ia, ia / MFMA_F64_PEAK_TFLOPS
                out["roofline_trailing_update"] = tu
        if timed_launches and ph["kbuild"] == ph["kbuild"]:
            # SE-kernel build: lower 64x64 tiles of K written once (+ X read), HIP events around its launch
            nbytes = (npad // 64) * (npad // 64 + 1) // 2 * 64 * 64 * 8 + args.n * args.d * 8
            kb_ms = ph["kbuild"]                      # phase events around the launch (levels < 5) ...
            if kb_st and kb_st["launches"] > 0:       # ... or the launch's own workgroup stamps, averaged over the pass
                kb_ms = kb_st["sum_ms"] / kb_st["launches"]
            gbs = nbytes / (kb_ms * 1e-3) / 1e9
            out["roofline_kbuild"] = {"kernel": "k_build (SE covariance, lower tiles)", "bound": "hbm", "achieved": gbs,
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                      "traffic": pmc.get("k_build", {}).get("hbm_bytes_per_launch"),
                                      "traffic_source": PMC_SUMMARY if "k_build" in pmc else (pmc_reason or "kernel not in %s" % PMC_SUMMARY),
                                      "launch_us": 1e3 * kb_ms, "algorithmic_bytes": nbytes,
                                      "valu_issue_frac_pmc": pmc.get("k_build", {}).get("valu_issue_frac"),
                                      "valu_per_wave_pmc": pmc.get("k_build", {}).get("valu_per_wave"),
                                      "note": "bound by VALU issue, not by HBM or its stores (valu_issue_frac_pmc = wave64 vector "
                                              "instructions x 4 cycles / 1024 SIMDs over the launch's cycles; per matrix entry 30 fp64 "
                                              "ops of the squared distance, contraction off to match the reference bit for bit, plus "
                                              "the exponential): rocprofv3 --pmc SQ_INSTS_VALU ... SQ_VMEM_WR_TA_DATA_FIFO_FULL"}
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(min(args.cpu_sample, args.n), args.n, args.d)
        sys.stdout.flush()
        os.dup2(fd_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)

    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
