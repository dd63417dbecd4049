"""Stand-alone dense-LA timings: Cholesky / triangular inverse / K^-1 at the sizes the
reference's library probes used (potrf N=7000, 32000; trsm 1000; gemm 4096) plus the metric size."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cugp_amd import capi                                  # noqa: E402

sizes = [int(s) for s in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1000,4096,7000,8192,16384,32000".split(","))]
names = {0: "potrf", 1: "trtri", 2: "lauum(K^-1)", 3: "potrf+trtri+lauum", 4: "gemm n^3"}
for n in sizes:
    row = []
    for op in (0, 1, 2, 3, 4):
        ms = C.c_double()
        capi.check(capi.lib().cugp_bench_la(op, n, 0, 3, C.byref(ms)))
        npad = -(-n // 128) * 128
        flop = 2.0 * npad ** 3 if op == 4 else (n ** 3 / 3.0) * (3 if op == 3 else 1)
        row.append("%s %.3f ms (%.1f TF/s)" % (names[op], ms.value, flop / ms.value / 1e9))
    print("n=%6d  " % n + "  ".join(row), flush=True)
