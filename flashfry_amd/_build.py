"""Build helpers: compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "flashfry_amd", "csrc")
LIB_DIR = os.path.join(ROOT, "flashfry_amd", "lib")
LIB = os.path.join(LIB_DIR, "libflashfry_hip.so")

HIP_SOURCES = ["ffh_api.hip"]
CXX_SOURCES = ["ffh_dbfile.cpp", "ffh_dbwrite.cpp"]
DEPS = ["ffh_api.hip", "ffh_ctx.hpp", "ffh_devbuf.hpp", "ffh_abi_guard.hpp", "ffh_plan.inc", "ffh_context.inc", "ffh_load.inc", "ffh_scan.inc", "ffh_finalize.inc", "ffh_pipe.inc", "ffh_share.inc", "ffh_index_api.inc", "ffh_bulge_api.inc", "ffh_exchange.inc", "ffh_debug.hpp", "ffh_streams.hpp", "ffh_kernels.hpp", "ffh_compare.hpp", "ffh_prims.hpp", "ffh_dbfile.cpp", "ffh_dbfile.hpp", "ffh_dbwrite.cpp", "ffh_ingest.hpp", "ffh_index.hpp", "ffh_inflate.hpp", "ffh_bulge.hpp", "ffh_comm.hpp", "ffh_exchange_kernels.hpp", "cfd_table.inc", "jost_table.inc",
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
           "-Wall", "-Wno-unused-result", "-I/opt/rocm/include", "-o", LIB]
    cmd += [os.path.join(CSRC, s) for s in HIP_SOURCES + CXX_SOURCES]
    cmd += ["-lz", "-lpthread", "-ldl"]   # (librccl is opened with dlopen when a communicator is created: ffh_comm.hpp)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


HOST = os.path.join(ROOT, "flashfry_amd", "host")
BIN_DIR = os.path.join(ROOT, "flashfry_amd", "bin")
CLI = os.path.join(BIN_DIR, "flashfry-hip")
HOST_SOURCES = ["ffhost_core.cpp", "ffhost_table.cpp", "ffhost_index.cpp", "ffhost_cli.cpp"]


def build_cli(force=False, verbose=False):
    """the C++ host CLI (index / discover / score) on top of the C ABI; links libflashfry_hip.so by rpath"""
    build_hip_library(force=False, verbose=verbose)
    os.makedirs(BIN_DIR, exist_ok=True)
    deps = [os.path.join(HOST, f) for f in HOST_SOURCES + ["ffhost.hpp"]] + [LIB, os.path.join(ROOT, "include", "flashfry_hip.h")]
    if not force and not stale(CLI, deps):
        return CLI
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-ffp-contract=off", "-o", CLI] + [os.path.join(HOST, f) for f in HOST_SOURCES]
    cmd += ["-L" + LIB_DIR, "-lflashfry_hip", "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath," + LIB_DIR, "-lz", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    print(build_cli(force=True, verbose=True))
    print(build_hip_library(force=True, verbose=True))
