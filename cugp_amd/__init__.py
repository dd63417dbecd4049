"""cugp_amd -- MI355X-native GP-regression hot path (drop-in for cuGP's Covsum / BCM objective).

The product is cugp_amd/lib/libcugp.so (C-ABI: include/cugp.h; kernels: cugp_amd/csrc/).
`cugp_amd.gp` mirrors the reference's host classes over that ABI; `cugp_amd.bcm` shards experts
one process per GPU with an RCCL all-reduce.  Nothing here computes on the CPU.
"""
import os as _os

_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # see csrc/cugp_capi.cpp (QueueDefault); no effect once HIP is up

from .gp import BCM, Covsum, cg_minimize, poe_finish, rprop_minimize  # noqa: F401

__all__ = ["Covsum", "BCM", "cg_minimize", "rprop_minimize", "poe_finish"]
