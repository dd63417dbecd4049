"""Build libcugp.so (HIP kernels + C-ABI) for gfx950, in tree.

    python -m cugp_amd.build           # rebuild if sources are newer than the library

hipcc cross-compiles without a GPU; the resulting cugp_amd/lib/libcugp.so travels with the tree.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcugp.so")
SOURCES = ["kernels.hip", "cugp_capi.cpp", "bcm.cpp", "minimize.cpp", "comm.cpp"]
HEADERS = ["kernels.h", "group.h"]
PUBLIC_HEADER = os.path.join(os.path.dirname(HERE), "include", "cugp.h")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-result",
         "-x", "hip"]
LINK = ["-ldl"]


IDFILE = os.path.join(LIBDIR, "libcugp.id")


def source_hash():
    """sha256 (first 16 hex digits) over the sources the library is built from, in a fixed order.  It is compiled into
    the library (cugp_build_id) and written into every measurement that is kept as a file (tools/pmc_summary.py), so
    bench.py can tell whether a committed counter summary belongs to the library it is timing."""
    h = hashlib.sha256()
    for f in [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [PUBLIC_HEADER]:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def stale():
    if not os.path.exists(LIB):
        return True
    if os.path.exists(IDFILE):                    # the id the library was built from, beside it (mtimes do not survive a checkout)
        with open(IDFILE) as f:
            return f.read().strip() != source_hash()
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [PUBLIC_HEADER, os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    sid = source_hash()
    cmd = [HIPCC] + FLAGS + ['-DCUGP_BUILD_ID="%s"' % sid] + [os.path.join(CSRC, f) for f in SOURCES] + LINK + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(IDFILE, "w") as f:
        f.write(sid + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print("built", LIB)
