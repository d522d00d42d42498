"""Builds the native parts of msdfgen_amd IN-TREE with hipcc (gfx950 only; cross-compiles without a GPU).

    msdfgen_amd/lib/libmsdfgen_hip.so        HIP kernels + the C ABI of include/msdfgen_hip.h      (always)
    msdfgen_amd/lib/libmsdfgen_hip_shim.so   C++ drop-in with msdfgen's own generate*() signatures  (only where the msdfgen
                                             headers are available: MSDFGEN_INCLUDE or /root/reference)
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libmsdfgen_hip.so")
SHIM = os.path.join(LIBDIR, "libmsdfgen_hip_shim.so")

# -ffp-contract=off: the reference is compiled without FMA contraction (x86-64 baseline); parity needs the same on gfx950.
# -disable-machine-licm: gfx950 has no 64-bit literal operands, so every fp64 constant (the polynomial coefficients of acos / cos /
#   cbrt in the quadratic solver: ~45 of them) is materialised with two moves; MachineLICM hoists all of those out of the per-edge
#   loop and pins ~90 VGPRs for the whole kernel (k_distance<msdf>: 177 -> 95 VGPRs without it, i.e. 2 -> 5 wavefronts per SIMD
#   and no scratch spills in the overlapping-combiner instantiation). Re-materialising them next to their use is mostly SALU work.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-disable-machine-licm", "-fPIC", "-shared", "-Wall",
               "-Wno-unused-value"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def source_hash():
    """sha256 (12 hex digits) over the kernel / C-ABI sources: profiles record it, bench.py quotes a committed PMC profile only if it matches."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:12]


def build_lib(force=False, verbose=False):
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(ROOT, "include", "msdfgen_hip.h")]
    if not force and not _stale(LIB, srcs):
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc()] + HIPCC_FLAGS + [os.path.join(CSRC, "msdf_capi.hip"), "-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


def msdfgen_include_dir():
    for cand in (os.environ.get("MSDFGEN_INCLUDE"), "/root/reference"):
        if cand and os.path.exists(os.path.join(cand, "msdfgen.h")):
            return cand
    return None


def build_shim(force=False, verbose=False):
    """The C++ drop-in (msdfgen_amd/shim/msdfgen_shim.cpp) needs msdfgen's public headers; it is built against the user's msdfgen
    checkout (here: /root/reference, read in place -- nothing is copied)."""
    inc = msdfgen_include_dir()
    src = os.path.join(PKG, "shim", "msdfgen_shim.cpp")
    if inc is None or not os.path.exists(src):
        return None
    if not force and not _stale(SHIM, [src, LIB]):
        return SHIM
    cmd = ["g++", "-O2", "-std=c++11", "-fPIC", "-shared", "-DMSDFGEN_PUBLIC=", "-DMSDFGEN_USE_CPP11", "-I", inc, "-I", os.path.join(ROOT, "include"),
           src, "-o", SHIM, "-L", LIBDIR, "-lmsdfgen_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return SHIM


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
    print(build_shim(force="--force" in sys.argv, verbose=True))
