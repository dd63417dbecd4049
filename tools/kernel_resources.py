"""Registers, LDS and scratch of every kernel of libcugp.so, from the code-object notes hipcc writes for gfx950
(no GPU needed):  python tools/kernel_resources.py > profiles/r03_kernel_resources.txt
Workgroups per CU = min over the limits: 512 unified VGPRs per SIMD lane (VGPR + AGPR, allocation granule 8),
160 KiB LDS per CU (static + the dynamic size the launcher asks for), 32 waves per CU."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cugp_amd", "csrc", "kernels.hip")
# dynamic LDS per launch (kernels.hip: set_big_lds / the launchers): Geo<4>::LDS, Geo<2>::LDS, POTF2_LDS, TRTRI_LDS
GEMM4, GEMM2, POTF2, TRTRI = 66048, 33280, 79360, 78336
DYN = {"k_lauum<4>": GEMM4, "k_lauum<2>": GEMM2, "k_trtri_level<4>": GEMM4, "k_trtri_level<2>": GEMM2,
       "k_trtri_border<4>": GEMM4, "k_trtri_border<2>": GEMM2, "k_predict_gemm": GEMM4, "k_test_gemm": GEMM4,
       "k_syrk_wide": GEMM4, "k_syrk_step": POTF2, "k_potf2": POTF2, "k_trtri_diag": TRTRI}

with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-S",
                           "--cuda-device-only", "-o", out, SRC], stderr=subprocess.DEVNULL)
    txt = open(out).read()
blocks = re.split(r"\n  - \.agpr_count:", txt[txt.index("amdhsa.kernels:"):])
print("%-22s %5s %5s %5s %8s %8s %8s %7s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "lds_stat", "lds_dyn", "scratch", "threads", "wg/CU"))
for b in blocks[1:]:
    def f(key):
        m = re.search(r"\.%s:\s+(\S+)" % key, b)
        return m.group(1) if m else "0"
    agpr = int(b.strip().split()[0])
    name = f("name")
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"^void |cugp::|\(.*$", "", dem)
    vg, sg = int(f("vgpr_count")), int(f("sgpr_count"))
    lds, scr, thr = int(f("group_segment_fixed_size")), int(f("private_segment_fixed_size")), int(f("max_flat_workgroup_size"))
    dyn = DYN.get(short, 0)
    waves = thr // 64
    regs = -(-(-(-vg // 4) * 4 + agpr) // 8) * 8   # unified file: accumulation registers follow the (4-aligned) VGPRs
    by_reg = (512 // regs) * 4 // waves if regs else 99
    by_lds = (160 * 1024) // (lds + dyn) if lds + dyn else 99
    by_waves = 32 // waves
    print("%-22s %5d %5d %5d %8d %8d %8d %7d %6d" % (short, vg, agpr, sg, lds, dyn, scr, thr, min(by_reg, by_lds, by_waves)))
