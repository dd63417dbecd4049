"""One evaluation through libcugp.so, THEN the first torch device call, in one process (mode cugp_first; torch_first:
the other order): both must work whichever HIP runtime copy got loaded first (cugp_amd/capi.py)."""
import os
import sys

import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
mode = sys.argv[1]
if mode == "torch_first":
    import torch
import cugp_amd.gp as gp
from conftest import synth
X, y = synth(300, 4, seed=1)
g = gp.Covsum(300, 4); g.set_data(X, y); g.set_loghyperparam(np.array([1.0, 0.2, -1.0]))
print("ll", g.loglik_grad()[0], flush=True)
import torch
t = torch.zeros(4, device="cuda:0", dtype=torch.float64)
print(mode, "torch ok", t.sum().item(), flush=True)

print([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l][::8])
