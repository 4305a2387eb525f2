"""Builds libmuse_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

The shared library is kept next to this file so that it travels with a repository snapshot to a
GPU box (it is git-ignored, not gpurun-ignored).  hipcc cross-compiles without a GPU present.
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libmuse_hip.so")
SOURCES = [os.path.join(CSRC, "muse_engine.hip"), os.path.join(CSRC, "muse_comm.cpp")]
HEADERS = [os.path.join(CSRC, h) for h in ("rng.hpp", "args.hpp", "vec.hpp", "reduce.hpp", "models.hpp", "solver.hpp")] + \
    [os.path.join(_HERE, "..", "include", "muse_hip.h")]
# -ffp-contract=off: the sampler's log/sincos sequences and the model gradients are defined in terms
# of individually rounded IEEE operations (bit-equal to a host evaluation of the same sequence).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wno-unused-value"]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in SOURCES + HEADERS)


def build_extension(force=False, verbose=False):
    """Compile the HIP engine; returns the path of the shared library."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libmuse_hip.so")
    tmp = LIB_PATH + ".tmp"
    cmd = [hipcc] + HIPCC_FLAGS + SOURCES + ["-o", tmp, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))
