"""bench.py's per-kernel records against rocprofv3: for every kernel of `roofline_kernels` in a bench line,
algorithmic flop per launch / AverageNs of the kernel-trace statistics of the TIMED pass (the default path `value` is
measured on) next to the bench's own `achieved` (profiled pass, launches stamped by their own workgroups).
    python tools/roofline_check.py [profiles/r05_bench_line.json] [profiles/r05_bench_timed_n8192_kernel_stats.csv]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
line = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_bench_line.json")
stats = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r05_bench_timed_n8192_kernel_stats.csv")
d = json.loads(open(line).read().strip().splitlines()[-1])
rows = {r["Name"]: r for r in csv.DictReader(open(stats))}
print("%s  vs  %s" % (os.path.relpath(line, ROOT), os.path.relpath(stats, ROOT)))
print("%-20s %12s %10s %14s %14s %8s" % ("kernel", "flop/launch", "AverageNs", "csv TF/s", "bench TF/s", "ratio"))
for k, v in d["roofline_kernels"].items():
    m = [r for n, r in rows.items() if k in n]
    if not m:
        continue
    ns = float(m[0]["AverageNs"])
    c = v["algorithmic_flop_per_launch"] / ns * 1e-3
    print("%-20s %12.4g %10.0f %14.2f %14.2f %8.3f%s" % (k, v["algorithmic_flop_per_launch"], ns, c, v["achieved"], v["achieved"] / c,
                                                        "   <- roofline (dominant kernel)" if k in d["roofline"]["kernel"] else ""))
tu = d.get("roofline_trailing_update")
if tu:
    ks, kw = d["roofline_kernels"]["k_syrk_step"], d["roofline_kernels"]["k_syrk_wide"]
    ns = {k: float([r for n, r in rows.items() if k in n][0]["AverageNs"]) for k in ("k_syrk_step", "k_syrk_wide")}
    per_eval = lambda v: v["launches_timed"] / d["steps"]
    fl = ks["algorithmic_flop_per_launch"] * per_eval(ks) + kw["algorithmic_flop_per_launch"] * per_eval(kw)
    t = ns["k_syrk_step"] * per_eval(ks) + ns["k_syrk_wide"] * per_eval(kw)
    print("trailing update (k_syrk_step + k_syrk_wide): csv %.2f TF/s = %.3f of %.1f; bench %.2f TF/s = %.3f"
          % (fl / t * 1e-3, fl / t * 1e-3 / tu["peak"], tu["peak"], tu["achieved"], tu["frac"]))
