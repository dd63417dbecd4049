#!/bin/bash
# Board power and sclk while (a) the plain fp64 tile product loops (70 TF/s), (b) the metric evaluation loops, (c) idle:
# is the evaluation's lower in-kernel clock (2.13 GHz, tools/wgtimes.hip) a power cap?   tools/power_probe.sh > out.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
sample() { for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Power (W)\|sclk" | tr '\n' ' '; echo; sleep 0.4; done; }
echo "== power cap / limits"; rocm-smi --showmaxpower 2>/dev/null | grep -i "power" ; rocm-smi --showpowercap 2>/dev/null | grep -i cap
echo "== idle"; sample | head -2
echo "== plain fp64 tile product n=8192 (cugp_bench_la op 4), looping"
( python3 - <<'PY'
import ctypes as C, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cugp_amd import capi
ms = C.c_double()
for i in range(12):
    capi.check(capi.lib().cugp_bench_la(4, 8192, 0, 20, C.byref(ms)))
print("gemm %.3f ms per product" % ms.value)
PY
) & BP=$!; sleep 3; sample; wait $BP
echo "== metric evaluation (bench.py --passes timed), looping"
( python3 bench.py --passes timed --steps 500 --cpu-sample 0 --sub-steps 0 > /dev/null 2>&1 ) & BP=$!; sleep 5; sample; wait $BP
