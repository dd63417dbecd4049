"""Summarise rocprofv3 PMC passes (tools/pmc.sh, tools/make_profiles.sh) into profiles/<tag>_pmc_summary.json
(python tools/pmc_summary.py [gpurun_out dir] [tag]).
FETCH_SIZE is doubled before use: on gfx950 it reports half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-byte-per-lane stores.  Units: KiB.
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x shader clock x SIMDs of the chip): the counter is in
cycles summed over the SIMDs (MI355X_MICROARCH.md, row 's_memtime tick vs SQ PMC units'), the kernel time is the
dispatch's own begin..end in the same pass (counter passes serialise the dispatches: isolated durations)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
tag = sys.argv[2] if len(sys.argv) > 2 else "r06"
sys.path.insert(0, ROOT)
from cugp_amd import build as _build    # noqa: E402   (source_hash: the id compiled into the library that was profiled)
CLOCK_HZ, SIMDS = 2.4e9, 256 * 4
KERNELS = ("k_syrk_step", "k_syrk_wide", "k_lauum<4>", "k_lauum<2>", "k_trtri_level<4>", "k_trtri_level<2>",
           "k_trtri_border<4>", "k_trtri_border<2>", "k_trtri_diag", "k_trtri_block", "k_build", "k_trace", "k_potf2",
           "k_trsm_inv64")


def load(t):
    f = glob.glob(os.path.join(src, "pmc_%s" % t, "*", "*counter_collection.csv"))
    f.sort(key=os.path.getmtime)                       # gpurun_out keeps earlier runs: take the newest
    return list(csv.DictReader(open(f[-1]))) if f else []


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


def git_head():
    try:
        import subprocess
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return os.environ.get("CUGP_GIT_HEAD")          # (the GPU box has no .git: tools/make_profiles.sh may pass it)


# build_id: hash of the library's sources as they are in THIS tree (the library in the tree was built from them:
# cugp_amd/build.py rebuilds when its recorded id differs); bench.py reports roofline.traffic from this file only
# when the library it has loaded carries the same id
idfile = os.path.join(ROOT, "cugp_amd", "lib", "libcugp.id")
built = open(idfile).read().strip() if os.path.exists(idfile) else None
out = {"note": __doc__.strip(), "build_id": _build.source_hash(), "library_id_file": built, "git_head": git_head(),
       "kernels": {}}
if built and built != out["build_id"]:
    raise SystemExit("the library in the tree (%s) was not built from the sources in the tree (%s): rebuild, re-profile" % (built, out["build_id"]))
for t, key in (("FETCH_SIZE", "fetch_kib_raw"), ("WRITE_SIZE", "write_kib")):
    agg, n = collections.defaultdict(float), collections.Counter()
    for r in load(t):
        k = short(r["Kernel_Name"])
        if k and r["Counter_Name"] == t:
            agg[k] += float(r["Counter_Value"])
            n[k] += 1
    for k in agg:
        d = out["kernels"].setdefault(k, {})
        d[key] = agg[k]
        d["launches"] = n[k]
sq = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
seen = set()
for r in load("SQ_WAVE_CYCLES"):
    k = short(r["Kernel_Name"])
    if k:
        sq[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:               # one row per counter: count a dispatch's duration once
            seen.add(r["Dispatch_Id"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
for k, v in sq.items():
    d = out["kernels"].setdefault(k, {})
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    d["sq"] = {n: x for n, x in v.items()}
    d["kernel_time_s_in_counter_pass"] = dur[k]
    if dur[k] > 0:
        d["mfma_busy_frac"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (dur[k] * CLOCK_HZ * SIMDS)
    if wc:
        d["wait_inst_any_frac"] = v.get("SQ_WAIT_INST_ANY", 0.0) / wc
        d["lds_bank_conflict_frac"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / wc
# round 5: the VALU pass (what k_build / k_trace are bound by): a wave64 vector instruction holds its SIMD's issue for 4
# cycles (16 lanes per SIMD), so instructions x 4 / SIMDs against the launch's cycles is the share of the VALU issue used
vseen = set()
vdur = collections.defaultdict(float)
vsum = collections.defaultdict(lambda: collections.defaultdict(float))
for r in load("SQ_INSTS_VALU"):
    k = short(r["Kernel_Name"])
    if k:
        vsum[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in vseen:
            vseen.add(r["Dispatch_Id"])
            vdur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
for k, v in vsum.items():
    d = out["kernels"].setdefault(k, {})
    d["valu"] = {n: x for n, x in v.items()}
    if vdur[k] > 0 and v.get("SQ_INSTS_VALU"):
        d["valu_issue_frac"] = v["SQ_INSTS_VALU"] * 4.0 / SIMDS / (vdur[k] * CLOCK_HZ)
        d["valu_per_wave"] = v["SQ_INSTS_VALU"] / v["SQ_WAVES"] if v.get("SQ_WAVES") else None
for k, d in out["kernels"].items():
    if "fetch_kib_raw" in d and "write_kib" in d and d.get("launches"):
        d["hbm_bytes_per_launch"] = (2.0 * d["fetch_kib_raw"] + d["write_kib"]) * 1024.0 / d["launches"]
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_pmc_summary.json" % tag), "w"), indent=1)
for k, d in out["kernels"].items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in d.items() if a != "sq"})
