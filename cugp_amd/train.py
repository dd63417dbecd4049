"""Distributed BCM training driver -- the MI355X counterpart of cuda_scalingdist/main.cpp.

Reference CLI (harness.sh:53-59):  ./gp $HOSTNAME $MASTER $W numchunks N D inprefix labprefix
Here, one process per GPU on one node:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 \
        -m cugp_amd.train --numchunks 16 --rows 1500 --inputs <prefix> --labels <prefix> [--test-rows T]

Chunk i is <prefix>i.txt and goes to rank i mod W (cg_solver.cpp:93); hyper-parameters start at
{2,2,2} (main.cpp:298-301); cg_solve with the reference's 100-evaluation budget; rank 0 prints the
"PLEASE-SEE 3" line and the training time like the reference does (cg_solver.cpp:518, main.cpp:305).
"""
import argparse
import os
import time

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--numchunks", type=int, required=True)
    ap.add_argument("--rows", type=int, required=True, help="training rows per chunk (numtrain)")
    ap.add_argument("--inputs", required=True, help="input file prefix")
    ap.add_argument("--labels", required=True, help="label file prefix")
    ap.add_argument("--hp", type=float, nargs=3, default=[2.0, 2.0, 2.0])
    ap.add_argument("--budget", type=int, default=100)
    ap.add_argument("--test-rows", type=int, default=0, help="rows after --rows in chunk 0 used as a held-out set")
    ap.add_argument("--backend", default="nccl")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from . import dataset
    from .bcm import ShardedBCM, expert_owner
    from .gp import Covsum

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    mine = {k for k in range(args.numchunks) if expert_owner(k, world) == rank}
    shards = dataset.load_shards(args.inputs, args.labels, args.numchunks, rows=None, only=mine)
    experts = [None if s is None else (s[0][:args.rows], s[1][:args.rows]) for s in shards]
    bcm = ShardedBCM(experts, rank=rank, world=world, device=local)
    bcm.set_loghyper(args.hp)
    t0 = time.perf_counter()
    trace = bcm.cg_solve(args.budget)
    dt = time.perf_counter() - t0
    if rank == 0:
        print("\n\n PLEASE-SEE 3 : %f, %f, %f\n" % tuple(bcm.hp))
        print("TOTAL training time = %f  (%d evaluations, final -LL %.9g)" % (dt, trace.shape[0], trace[-1, 3]))
    if args.test_rows > 0:
        s0 = dataset.load_chunk("%s0.txt" % args.inputs, "%s0.txt" % args.labels)
        Xt, yt = s0[0][args.rows:args.rows + args.test_rows], s0[1][args.rows:args.rows + args.test_rows]
        m, v = bcm.predict(Xt)
        if rank == 0:
            print("NLPP = %.12g" % Covsum.get_negative_log_predprob(yt, m, v))
    bcm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
