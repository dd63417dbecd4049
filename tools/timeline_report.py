"""Timeline of the LAST evaluation in a rocprofv3 kernel trace (…_kernel_trace.csv): per-launch start/duration,
per-kernel totals, and how much of the span had 0 / 1 / 2+ kernels in flight.
python tools/timeline_report.py <kernel_trace.csv> [--launches]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = re.sub(r"^void |cugp::|\(.*$|<.*$", "", r["Kernel_Name"])
    def dim(k):
        return max(int(r.get(k, 1) or 1), 1)
    wgs = (dim("Grid_Size_X") * dim("Grid_Size_Y") * dim("Grid_Size_Z")) // (dim("Workgroup_Size_X") * dim("Workgroup_Size_Y") * dim("Workgroup_Size_Z"))
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?"), r["Kernel_Name"], wgs))
ev.sort()
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_build")]
ev = ev[starts[-1]:]
t0 = ev[0][0]
end = max(e[1] for e in ev)
print("evaluation span %.3f ms, %d launches" % ((end - t0) / 1e6, len(ev)))
tot = defaultdict(lambda: [0, 0.0])
for s, e, nme, q, full, wgs in ev:
    key = nme + ("<2>" if "<2>" in full or "Li2E" in full else "")
    tot[key][0] += 1
    tot[key][1] += (e - s) / 1e3
for k, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("  %-28s %4d launches  %9.1f us total  %8.1f us avg" % (k, c, us, us / c))
# concurrency profile
pts = []
for s, e, *_ in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
lvl = 0; last = t0; hist = defaultdict(float)
for t, d in pts:
    hist[min(lvl, 2)] += (t - last) / 1e6
    lvl += d; last = t
print("  in flight: none %.3f ms, one kernel %.3f ms, two or more %.3f ms" % (hist[0], hist[1], hist[2]))
# how full the chip can be: workgroups of the launches in flight (an upper bound on what is resident: a launch's last,
# partly empty round counts whole) against the 512 slots of two 256-thread tile workgroups per CU
pts = []
for s, e, _n, _q, _f, wgs in ev:
    pts.append((s, wgs)); pts.append((e, -wgs))
pts.sort()
lvl = 0; last = t0; fill = defaultdict(float)
for t, d in pts:
    b = 0 if lvl == 0 else (1 if lvl < 128 else (2 if lvl < 512 else 3))
    fill[b] += (t - last) / 1e6
    lvl += d; last = t
print("  workgroups of the launches in flight: none %.3f ms, < 128 %.3f ms, 128-511 %.3f ms, >= 512 %.3f ms"
      % (fill[0], fill[1], fill[2], fill[3]))
if "--launches" in sys.argv:
    for s, e, nme, q, full, wgs in ev:
        print("%10.1f us  +%8.1f us  q%-3s %-18s %6d wg" % ((s - t0) / 1e3, (e - s) / 1e3, q, nme, wgs))
