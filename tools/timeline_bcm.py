"""Three BCM evaluations (K experts x rows) for a rocprofv3 --kernel-trace run; feed to tools/timeline_report.py."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cugp_amd.gp as gp
from cugp_amd import capi
from conftest import synth
K, rows = int(sys.argv[1]), int(sys.argv[2])
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    capi.check(capi.lib().cugp_set_tuning(int(k), int(v)))
X, y = synth(K * rows, seed=5)
b = gp.BCM.split(X, y, K)
hp = np.array([np.log(3.0), 0.0, np.log(0.1)])
for it in range(3):
    b.set_BCM_log_hyperparam(hp + 1e-3 * it)
    print(b.loglik_grad()[0], flush=True)
b.close()
