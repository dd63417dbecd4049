"""Total HBM-side traffic of the evaluations in one rocprofv3 --pmc pass pair (FETCH_SIZE, WRITE_SIZE):
python tools/pmc_total.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <number of evaluations in the run>
FETCH_SIZE is doubled (gfx950: half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section)."""
import collections
import csv
import glob
import os
import sys

src, nev = sys.argv[1], float(sys.argv[2])
tot = {}
per = collections.defaultdict(lambda: [0.0, 0.0, 0])
for i, t in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    f = sorted(glob.glob(os.path.join(src, "pmc_%s" % t, "*", "*counter_collection.csv")), key=os.path.getmtime)
    s = 0.0
    for r in csv.DictReader(open(f[-1])):
        if r["Counter_Name"] == t:
            v = float(r["Counter_Value"]) * 1024.0 * (2.0 if t == "FETCH_SIZE" else 1.0)
            s += v
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cugp::", "")
            per[k][i] += v
            if i == 0:
                per[k][2] += 1
    tot[t] = s
print("per evaluation: read %.2f GB, written %.2f GB, together %.2f GB" % (tot["FETCH_SIZE"] / nev / 1e9, tot["WRITE_SIZE"] / nev / 1e9,
                                                                           (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / nev / 1e9))
for k, v in sorted(per.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
    print("  %-28s %6d launches  read %8.3f GB  written %8.3f GB per evaluation" % (k, v[2] / nev, v[0] / nev / 1e9, v[1] / nev / 1e9))
