"""Build libcugp.so (HIP kernels + C-ABI) for gfx950, in tree.

    python -m cugp_amd.build           # rebuild if sources are newer than the library

hipcc cross-compiles without a GPU; the resulting cugp_amd/lib/libcugp.so travels with the tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcugp.so")
SOURCES = ["kernels.hip", "cugp_capi.cpp", "bcm.cpp", "minimize.cpp"]
HEADERS = ["kernels.h", "group.h"]
PUBLIC_HEADER = os.path.join(os.path.dirname(HERE), "include", "cugp.h")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-result",
         "-x", "hip"]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [PUBLIC_HEADER, os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [HIPCC] + FLAGS + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print("built", LIB)
