"""Build helpers: compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "flashfry_amd", "csrc")
LIB_DIR = os.path.join(ROOT, "flashfry_amd", "lib")
LIB = os.path.join(LIB_DIR, "libflashfry_hip.so")

HIP_SOURCES = ["ffh_api.hip"]
CXX_SOURCES = ["ffh_dbfile.cpp"]
DEPS = ["ffh_api.hip", "ffh_kernels.hpp", "ffh_prims.hpp", "ffh_dbfile.cpp", "ffh_dbfile.hpp", "cfd_table.inc",
        os.path.join("..", "..", "include", "flashfry_hip.h")]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")


def stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_hip_library(force=False, verbose=False):
    os.makedirs(LIB_DIR, exist_ok=True)
    deps = [os.path.join(CSRC, d) for d in DEPS]
    if not force and not stale(LIB, deps):
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-Wall", "-Wno-unused-result", "-o", LIB]
    cmd += [os.path.join(CSRC, s) for s in HIP_SOURCES + CXX_SOURCES]
    cmd += ["-lz", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_hip_library(force=True, verbose=True))
