"""Builds libmuse_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

The shared library is kept next to this file so that it travels with a repository snapshot to a
GPU box (it is git-ignored, not gpurun-ignored).  hipcc cross-compiles without a GPU present.
Translation units, each compiled to an object of its own and rebuilt only when it (or a header it includes) changed:
the device code -- muse_kernels.hip (dispatch, per-simulation operator kernels) and kernels_part.hip once per group of
models (every model x placement instantiation of the solver and loop kernels; the groups compile side by side) --,
muse_engine.cpp the host side and the C ABI (seconds), muse_comm.cpp the exchange between ranks (RCCL, shared memory).
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libmuse_hip.so")
OBJ_DIR = os.path.join(_HERE, "build")
_API = os.path.join(_HERE, "..", "include", "muse_hip.h")
_KERNEL_HEADERS = [os.path.join(CSRC, h) for h in ("rng.hpp", "args.hpp", "vec.hpp", "reduce.hpp", "models.hpp", "user_model.hpp", "solver.hpp",
                                                    "step.hpp", "kernels.hpp")]
_SWITCHES = os.path.join(CSRC, "switches.hpp")
# source -> (headers it depends on, extra flags)
# -ffp-contract=off: the sampler's log/sincos sequences and the model gradients are defined in terms
# of individually rounded IEEE operations (bit-equal to a host evaluation of the same sequence).
_DEVICE_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off"]
KERNEL_PARTS = 8   # kernels.hpp: MUSE_PART_0 .. 7 (the groups of models whose solver / loop kernels one unit instantiates)
# object name -> (source, headers it depends on, extra flags).  Device code: muse_kernels.hip (dispatch, per-simulation operator kernels)
# and kernels_part.hip once per group of models (round 5: ONE unit with every instantiation took 6.5 minutes; the groups compile side
# by side in ~1.5).
UNITS = {
    "muse_kernels": ("muse_kernels.hip", _KERNEL_HEADERS + [_API], _DEVICE_FLAGS),
    **{f"kernels_part{n}": ("kernels_part.hip", _KERNEL_HEADERS + [_API], _DEVICE_FLAGS + [f"-DMUSE_PART={n}"]) for n in range(KERNEL_PARTS)},
    # host code: plain C++ against the HIP runtime API (no device pass)
    # (-ffp-contract=off here too: step.hpp's algebra must round on the host exactly as in the step kernel)
    "muse_engine": ("muse_engine.cpp", [os.path.join(CSRC, "args.hpp"), os.path.join(CSRC, "step.hpp"), os.path.join(CSRC, "user_model.hpp"), _SWITCHES, _API],
                        ["-x", "c++", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-O2", "-ffp-contract=off"]),
    # (step.hpp's algebra again: the sharded muse! loop takes the same step as muse_run)
    "muse_comm": ("muse_comm.cpp", [_API, os.path.join(CSRC, "shm_gather.hpp"), os.path.join(CSRC, "args.hpp"), os.path.join(CSRC, "step.hpp"), _SWITCHES],
                      ["-x", "c++", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-O2", "-ffp-contract=off"]),
}
COMMON_FLAGS = ["-std=c++17", "-fPIC", "-Wno-unused-value"]
SOURCES = sorted({os.path.join(CSRC, u[0]) for u in UNITS.values()})
HEADERS = _KERNEL_HEADERS + [_API, os.path.join(CSRC, "shm_gather.hpp"), _SWITCHES]


def declared_symbols():
    """The entry points include/muse_hip.h declares (comments stripped): the library's export list."""
    import re
    text = re.sub(r"/\*.*?\*/", "", open(_API).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(muse_[a-z_A-Z0-9]+)\s*\(", text)))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in deps)


class _BuildLock:
    """One builder per library at a time, across processes (the ranks of a job all ask for the same model's library on first
    use): an flock on <library>.lock; whoever comes second finds the library fresh."""

    def __init__(self, lib_path):
        self.path = lib_path + ".lock"

    def __enter__(self):
        import fcntl
        self.f = open(self.path, "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()
        return False


def needs_build():
    return _stale(LIB_PATH, SOURCES + HEADERS)


def build_extension(force=False, verbose=False, defines=(), lib_path=None):
    """Compile the HIP engine; returns the path of the shared library.  `defines` (e.g. ["-DMUSE_STAMPS"]) and
    `lib_path` build a diagnostic variant next to the product library (objects are not cached for those)."""
    lib_path = lib_path or LIB_PATH
    variant = bool(defines) or lib_path != LIB_PATH
    if not force and not variant and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libmuse_hip.so")
    # a variant (diagnostic build, a user model's library) keeps its objects in a directory of its own: two of them may build at once
    # (next to the library it builds: with MUSE_MODEL_DIR the package directory may be read-only)
    obj_dir = (os.path.join(os.path.dirname(os.path.abspath(lib_path)), "build_" + os.path.splitext(os.path.basename(lib_path))[0])
               if variant else OBJ_DIR)
    os.makedirs(obj_dir, exist_ok=True)
    objs, todo = [], []
    for name, (src, deps, flags) in UNITS.items():
        path = os.path.join(CSRC, src)
        obj = os.path.join(obj_dir, name + ".o")
        objs.append(obj)
        if force or variant or _stale(obj, [path] + deps):
            todo.append([hipcc] + COMMON_FLAGS + flags + list(defines) + ["-c", path, "-o", obj])
    # the units compile side by side, MUSE_BUILD_JOBS (default: the CPUs this process may use) at a time
    jobs = int(os.environ.get("MUSE_BUILD_JOBS", "0")) or len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 4)
    running = []
    while todo or running:
        while todo and len(running) < max(1, jobs):
            cmd = todo.pop(0)
            if verbose:
                print(" ".join(cmd))
            running.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
        cmd, p = running.pop(0)
        if p.wait() != 0:
            for _, q in running:
                q.kill()
            raise subprocess.CalledProcessError(p.returncode, cmd)
    tmp = lib_path + ".tmp"
    # the library exports exactly what include/muse_hip.h declares: a linker version script written from the header (the accessors
    # between muse_engine.cpp and muse_comm.cpp, the kernels' host stubs and the C++ runtime's weak symbols stay local)
    vmap = os.path.join(obj_dir, "exports.map")
    with open(vmap, "w") as f:
        f.write("{\n  global:\n" + "".join(f"    {n};\n" for n in declared_symbols()) + "  local: *;\n};\n")
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp, "-ldl", "-lrt", "-lpthread", f"-Wl,--version-script={vmap}"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link, cwd=CSRC)
    os.replace(tmp, lib_path)
    return lib_path


# ------------------------------------------------------------------------------------------------------------------
# User-supplied elementwise models (include/muse_model.h; SimpleMuseProblem's closures, src/simple.jl:79-95, as compiled
# code): the same three translation units with -DMUSE_USER_MODEL_HEADER="<header>" give an engine library of their own
# that holds that one model (model id MUSE_MODEL_USER) behind the whole C ABI.
INCLUDE_DIR = os.path.normpath(os.path.join(_HERE, "..", "include"))
MODELS_DIR = os.path.join(_HERE, "models")


def model_out_dir():
    """Where model libraries (and the headers from_source writes) go: next to libmuse_hip.so, so that they travel with a
    snapshot of the repository -- or MUSE_MODEL_DIR (an installed, read-only package)."""
    d = os.environ.get("MUSE_MODEL_DIR")
    if d:
        os.makedirs(d, exist_ok=True)
        return os.path.abspath(d)
    return _HERE


def model_lib_path(name):
    return os.path.join(model_out_dir(), f"libmuse_hip_model_{name}.so")


def build_model_library(header, name, force=False, verbose=False):
    """Compile (if missing or stale) the engine library of the model in `header`; returns its path (in-tree, next to
    libmuse_hip.so, so that it travels with a snapshot of the repository)."""
    header = os.path.abspath(header)
    if not os.path.exists(header):
        raise FileNotFoundError(header)
    path = model_lib_path(name)
    deps = SOURCES + HEADERS + [header, os.path.join(INCLUDE_DIR, "muse_model.h")]
    if force or _stale(path, deps):
        with _BuildLock(path):
            if force or _stale(path, deps):     # (another process may have built it while this one waited)
                build_extension(defines=[f'-DMUSE_USER_MODEL_HEADER="{header}"', "-I" + INCLUDE_DIR], lib_path=path, verbose=verbose)
    return path


def build_packaged_models(force=False):
    """The example models shipped in museinference.jl_amd/models/*.h -> {name: library path}."""
    out = {}
    for f in sorted(os.listdir(MODELS_DIR)):
        if f.endswith(".h"):
            out[f[:-2]] = build_model_library(os.path.join(MODELS_DIR, f), f[:-2], force=force)
    return out


if __name__ == "__main__":
    import sys
    if "--stamps" in sys.argv:  # the -DMUSE_STAMPS diagnostic build read by tools/stamps.py
        print(build_extension(force=True, verbose=True, defines=["-DMUSE_STAMPS"],
                              lib_path=os.path.join(_HERE, "libmuse_hip_stamps.so")))
    else:
        print(build_extension(force=True, verbose=True))
