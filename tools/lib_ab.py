"""Two builds of libcugp.so on the same board: tools/lib_ab.py <lib A> <lib B> [n ...] -- stand-alone LA timings
(cugp_bench_la ops 0 = Cholesky, 3 = factorisation + inverse as an evaluation runs them), alternating A, B, A, B
in child processes (a process can load only one of them)."""
import ctypes as C
import subprocess
import sys

if sys.argv[1] == "--child":
    L = C.CDLL(sys.argv[2])
    L.cugp_bench_la.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    out = []
    for n in [int(v) for v in sys.argv[3:]]:
        for op in (0, 3):
            ms = C.c_double()
            rc = L.cugp_bench_la(op, n, 0, 5, C.byref(ms))
            out.append("n=%d op%d %.3f ms%s" % (n, op, ms.value, "" if rc == 0 else " rc=%d" % rc))
    print("  ".join(out), flush=True)
else:
    sizes = sys.argv[3:] or ["1536", "4096", "8192"]
    for rnd in range(2):
        for lib in sys.argv[1:3]:
            r = subprocess.run([sys.executable, __file__, "--child", lib] + sizes, capture_output=True, text=True)
            print("%-28s %s" % (lib.split("/")[-1], r.stdout.strip() or r.stderr.strip()[-300:]), flush=True)
