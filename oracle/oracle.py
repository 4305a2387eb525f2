"""ctypes binding of the CPU oracle (oracle/muse_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (museinference.jl_amd/) never does.  See muse_oracle.c for the reference
file:line each entry follows and for the parity status ("parity unpinned" vs reference outputs,
pinned against closed forms / scipy / Philox known-answer vectors).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmuse_oracle.so")

MODEL_FUNNEL, MODEL_NOISE, MODEL_SMOOTH = 0, 1, 2
MODELS = {"funnel": MODEL_FUNNEL, "noise": MODEL_NOISE, "smooth": MODEL_SMOOTH,
          "user": 3}   # "user": only inside `with user_model(header, name)` (a build of this file with that header compiled in)
STATUS = ["g_converged", "x_converged", "f_converged", "maxiter", "linesearch_failed", "nonfinite"]


class Info(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("f_calls", C.c_int32),
        ("status", C.c_int32),
        ("hist_words", C.c_int32),
        ("f_min", C.c_double),
        ("gnorm", C.c_double),
    ]


INFO_DTYPE = np.dtype(
    [("iterations", "<i4"), ("f_calls", "<i4"), ("status", "<i4"), ("hist_words", "<i4"),
     ("f_min", "<f8"), ("gnorm", "<f8")]
)


def build(force=False):
    """Compile libmuse_oracle.so with the committed Makefile (gcc)."""
    src = os.path.join(_HERE, "muse_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_selected = None   # the library of a user's model while `with user_model(...)` is active


def _declare(l):
    l.mo_negloglike_grad.restype = C.c_double
    l.mo_logLike_and_grad_z.restype = C.c_double
    l.mo_num_threads.restype = C.c_int
    l.mo_expd.restype = C.c_double
    l.mo_expd.argtypes = [C.c_double]
    l.mo_user_model_name.restype = C.c_char_p
    return l


def lib():
    global _lib
    if _selected is not None:
        return _selected
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = _declare(C.CDLL(_LIB_PATH))
    return _lib


def build_user_model(header, name, force=False):
    """The checker of a user-supplied model (include/muse_model.h): this oracle with the model's header compiled in as
    model "user" (Makefile target `user`) -> oracle/libmuse_oracle_model_<name>.so."""
    header = os.path.abspath(header)
    path = os.path.join(_HERE, f"libmuse_oracle_model_{name}.so")
    deps = [header, os.path.join(_HERE, "muse_oracle.c"), os.path.join(_HERE, "..", "include", "muse_model.h")]
    if force or not os.path.exists(path) or any(os.path.getmtime(path) < os.path.getmtime(d) for d in deps):
        subprocess.check_call(["make", "-C", _HERE, "-B", "user", f"NAME={name}", f"HEADER={header}"], stdout=subprocess.DEVNULL)
    return path


class fast_build:
    """with fast_build(): every function of this module runs in libmuse_oracle_fast.so -- the same source compiled for THIS
    machine with re-association and contraction allowed (Makefile target `fast`).  For bench.py's second CPU figure only:
    its results are compared with the strict build's at a tolerance, it is never the checker."""
    _lib = None

    def __init__(self, rebuild=True):
        path = os.path.join(_HERE, "libmuse_oracle_fast.so")
        if rebuild or not os.path.exists(path):   # (-march=native: built on the machine that runs it)
            subprocess.check_call(["make", "-C", _HERE, "-B", "fast"], stdout=subprocess.DEVNULL)
            fast_build._lib = None
        if fast_build._lib is None:
            fast_build._lib = _declare(C.CDLL(path))

    def __enter__(self):
        global _selected
        self._prev, _selected = _selected, fast_build._lib
        return self

    def __exit__(self, *exc):
        global _selected
        _selected = self._prev
        return False


class user_model:
    """with user_model(header, name): every function of this module runs in the build that holds that model (model "user");
    the built-in models are there too."""
    _cache = {}

    def __init__(self, header, name):
        self.path = build_user_model(header, name)

    def __enter__(self):
        global _selected
        if self.path not in self._cache:
            self._cache[self.path] = _declare(C.CDLL(self.path))
        self._prev, _selected = _selected, self._cache[self.path]
        return self

    def __exit__(self, *exc):
        global _selected
        _selected = self._prev
        return False


def set_constants(k, values):
    """Run-time constant vector k of the user model whose build is selected (`with user_model(...)`): include/muse_model.h,
    muse_const -- the checker's counterpart of muse_set_constants."""
    v = np.ascontiguousarray(np.asarray(values, dtype=np.float64)).reshape(-1)
    rc = lib().mo_set_constants(C.c_int(int(k)), v.ctypes.data_as(C.c_void_p), C.c_int64(v.size))
    if rc != 0:
        raise ValueError(f"mo_set_constants({k}) -> {rc}: the selected build's model declares no such constant")


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f8(a):
    return np.ascontiguousarray(np.atleast_1d(np.asarray(a, dtype=np.float64)))


def exp_fixed(x):
    """The fixed-sequence exponential behind the models' exp(theta/2), exp(-theta) (muse_oracle.c, mo_exp)."""
    return float(lib().mo_expd(float(x)))


def lbfgs_rosenbrock(g_tol=1e-8, x0=(0.0, 0.0)):
    """The restated Optim LBFGS()/HagerZhang() (mo_zhat_at_theta's solver) on Rosenbrock's function, the example of the
    Optim.jl documentation: returns (minimizer, iterations, f_calls, status, f_min, gnorm)."""
    z0 = np.array(x0, dtype=np.float64)
    dummy, th, zout, info = np.zeros(2), np.zeros(1), np.zeros(2), Info()
    rc = lib().mo_zhat_at_theta(C.c_int(100), C.c_int64(2), C.c_int(1), _p(dummy), _p(z0), _p(th), C.c_double(g_tol), _p(zout),
                                C.byref(info))
    assert rc == 0
    return zout, info.iterations, info.f_calls, info.status, info.f_min, info.gnorm


def philox4x32_10(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().mo_philox4x32_10(_p(c), _p(k), _p(out))
    return out


def normals(seed, sim, N):
    n1 = np.empty(N)
    n2 = np.empty(N)
    lib().mo_normals(C.c_uint64(seed), C.c_uint64(sim), C.c_int64(N), _p(n1), _p(n2))
    return n1, n2


def sample_x_z(model, N, seed, sim, theta):
    """sample_x_z(prob, rng, theta) -> (x, z)   [reference src/interface.jl:92-99]"""
    th = _f8(theta)
    x = np.empty(N)
    z = np.empty(N)
    lib().mo_sample_x_z(C.c_int(MODELS[model]), C.c_int64(N), C.c_int(th.size), C.c_uint64(seed),
                        C.c_uint64(sim), _p(th), _p(x), _p(z))
    return x, z


def logLike_and_grad_z(model, x, z, theta):
    """logLike_and_grad_z(prob,x,z,theta) -> (logLike, grad_z)   [src/interface.jl:68-83]"""
    th = _f8(theta)
    x = _f8(x)
    z = _f8(z)
    g = np.empty_like(z)
    f = lib().mo_logLike_and_grad_z(C.c_int(MODELS[model]), C.c_int64(x.size), C.c_int(th.size), _p(x),
                                    _p(z), _p(th), _p(g))
    return f, g


def grad_theta(model, x, z, theta):
    """grad_theta logLike(prob,x,z,theta)   [src/interface.jl:41-58]"""
    th = _f8(theta)
    x = _f8(x)
    z = _f8(z)
    out = np.empty(th.size)
    lib().mo_grad_theta(C.c_int(MODELS[model]), C.c_int64(x.size), C.c_int(th.size), _p(x), _p(z), _p(th),
                        _p(out))
    return out


def zhat_at_theta(model, x, z0, theta, atol=1e-2):
    """zhat_at_theta(prob,x,z0,theta; atol) -> (zhat, info)   [src/interface.jl:162-171]"""
    th = _f8(theta)
    x = _f8(x)
    z0 = _f8(z0)
    z = np.empty_like(x)
    info = Info()
    lib().mo_zhat_at_theta(C.c_int(MODELS[model]), C.c_int64(x.size), C.c_int(th.size), _p(x), _p(z0),
                           _p(th), C.c_double(atol), _p(z), C.byref(info))
    return z, {f: getattr(info, f) for f, _ in Info._fields_}


def map_and_score_batch(model, N, seed, sim_begin, sim_end, theta, atol=1e-2, x_data=None, z0_mode=0,
                        zhat=None, nthreads=1):
    """muse!/get_J! map body over a batch   [src/muse.jl:169-176, :508-525]"""
    th = _f8(theta)
    include_data = x_data is not None
    n = (sim_end - sim_begin) + (1 if include_data else 0)
    if zhat is None:
        zhat = np.zeros((n, N))
    assert zhat.shape == (n, N) and zhat.flags.c_contiguous
    g = np.empty((n, th.size))
    info = np.zeros(n, dtype=INFO_DTYPE)
    xd = _f8(x_data) if include_data else None
    lib().mo_map_and_score_batch(C.c_int(MODELS[model]), C.c_int64(N), C.c_int(th.size), C.c_uint64(seed),
                                 C.c_int64(sim_begin), C.c_int64(sim_end), C.c_int(int(include_data)),
                                 _p(xd), _p(th), C.c_double(atol), C.c_int(z0_mode), _p(zhat), _p(g),
                                 _p(info), C.c_int(nthreads))
    return g, zhat, info


def fd_jacobian(model, N, seed, sim, theta0, step, zfid, atol=1e-2):
    """get_H! finite-difference Jacobian for one sim   [src/muse.jl:426-433, src/util.jl:9-27]"""
    th = _f8(theta0)
    st = _f8(step)
    zf = _f8(zfid)
    H = np.empty((th.size, th.size))
    lib().mo_fd_jacobian(C.c_int(MODELS[model]), C.c_int64(N), C.c_int(th.size), C.c_uint64(seed),
                         C.c_int64(sim), _p(th), _p(st), C.c_double(atol), _p(zf), _p(H))
    return H


def implicit_H(model, N, seed, sim, theta0, atol=1e-1, cg_maxiter=100):
    """get_H! implicit-differentiation branch for one sim   [src/muse.jl:335-405]"""
    th = _f8(theta0)
    H = np.empty((th.size, th.size))
    its = np.zeros(th.size, dtype=np.int32)
    lib().mo_implicit_H(C.c_int(MODELS[model]), C.c_int64(N), C.c_int(th.size), C.c_uint64(seed), C.c_int64(sim),
                        _p(th), C.c_double(atol), C.c_int(cg_maxiter), _p(H), _p(its))
    return H, its


def num_threads():
    return lib().mo_num_threads()
