"""Three LL+grad evaluations at one size, no profiling events: run under `rocprofv3 --kernel-trace` and feed the
kernel trace to tools/timeline_report.py.   python3 tools/timeline_run.py <n> [pipe [key=value ...]]
TL_EXPERTS=K in the environment: a BCM of K experts over the n rows on one device instead of one expert."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cugp_amd.gp as gp                                  # noqa: E402
from cugp_amd import capi                                 # noqa: E402
from conftest import synth                                # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
if len(sys.argv) > 2:
    capi.check(capi.lib().cugp_set_tuning(3, int(sys.argv[2])))
for kv in sys.argv[3:]:                                   # further tuning keys as key=value (kernels.h TUNE_*)
    k, v = kv.split("=")
    capi.check(capi.lib().cugp_set_tuning(int(k), int(v)))
X, y = synth(n)
hp = np.array([np.log(3.0), 0.0, np.log(0.1)])
experts = int(os.environ.get("TL_EXPERTS", "0"))
if experts > 0:
    b = gp.BCM.split(X, y, experts)
    for it in range(3):
        b.set_BCM_log_hyperparam(hp + 1e-3 * it)
        print(b.loglik_grad()[:2], flush=True)
    b.close()
else:
    g = gp.Covsum(n, 10)
    g.set_data(X, y)
    for it in range(3):
        g.set_loghyperparam(hp + 1e-3 * it)
        print(g.loglik_grad(), flush=True)
    g.close()
