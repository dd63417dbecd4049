"""cugp_bench_la under `rocprofv3 --kernel-trace` (the report takes the last repetition: tools/timeline_report.py).
    python3 tools/timeline_la.py <op> <n> [key=value ...]        (kernels.h TUNE_* by number)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cugp_amd import capi                                  # noqa: E402

op, n = int(sys.argv[1]), int(sys.argv[2])
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    capi.check(capi.lib().cugp_set_tuning(int(k), int(v)))
ms = C.c_double()
capi.check(capi.lib().cugp_bench_la(op, n, 0, 2, C.byref(ms)))
print("op %d n %d: %.3f ms" % (op, n, ms.value), flush=True)
