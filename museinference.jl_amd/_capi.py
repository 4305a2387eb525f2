"""ctypes binding of libmuse_hip.so -- exactly the entry points declared in include/muse_hip.h.

There is no fallback: if the shared library is missing or no HIP device is usable, loading /
context creation raises.  Nothing in this package imports the CPU oracle (oracle/).
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

MODEL_FUNNEL, MODEL_NOISE, MODEL_SMOOTH, MODEL_USER = 0, 1, 2, 3
MODELS = {"funnel": MODEL_FUNNEL, "noise": MODEL_NOISE, "smooth": MODEL_SMOOTH}   # the built-in models of libmuse_hip.so
MEM_HOST, MEM_DEVICE = 0, 1
Z0_ZERO, Z0_TRUE, Z0_WARM = 0, 1, 2
MAX_THETA = 8
MAX_MAPS = 8
UNIQUE_ID_BYTES = 128
STATUS_NAMES = ("g_converged", "x_converged", "f_converged", "maxiter", "linesearch_failed", "nonfinite")
STATUS_MAXITER, STATUS_NONFINITE = 3, 5

INFO_DTYPE = np.dtype(
    [("iterations", "<i4"), ("f_calls", "<i4"), ("status", "<i4"), ("hist_words", "<i4"),
     ("f_min", "<f8"), ("gnorm", "<f8")]
)


class RunOptions(C.Structure):
    """struct muse_run_options of include/muse_hip.h"""
    _fields_ = [("nsims", C.c_int32), ("maxsteps", C.c_int32), ("theta_rtol", C.c_double), ("atol", C.c_double),
                ("alpha", C.c_double), ("prior_kind", C.c_int32), ("z0_warm", C.c_int32),
                ("prior_mean", C.c_double * MAX_THETA), ("prior_sigma", C.c_double * MAX_THETA)]


def run_hist_width(ntheta):
    return 7 * ntheta + ntheta * ntheta + 1  # MUSE_RUN_HIST


class MuseError(RuntimeError):
    """A libmuse_hip call returned a negative status."""

    def __init__(self, code, msg):
        super().__init__(f"libmuse_hip error {code}: {msg}")
        self.code = code


# symbol -> (restype, argtypes); keep in sync with include/muse_hip.h (tests check the export list)
_vp, _i, _i64, _u64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double
SIGNATURES = {
    "muse_ctx_create": (_i, [_i, _i64, _i, _i, C.POINTER(_vp)]),
    "muse_ctx_destroy": (_i, [_vp]),
    "muse_last_error": (C.c_char_p, []),
    "muse_model_name": (C.c_char_p, [_i]),
    "muse_set_data": (_i, [_vp, _vp, _i]),
    "muse_set_stream": (_i, [_vp, _vp]),
    "muse_set_placement": (_i, [_vp, _i]),
    "muse_max_resident_n": (_i64, []),
    "muse_set_element_split": (_i, [_vp, _i]),
    "muse_placement_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "muse_set_concurrency": (_i, [_vp, _i]),
    "muse_synchronize": (_i, [_vp]),
    "muse_last_kernel_ms": (_i, [_vp, C.POINTER(C.c_float)]),
    "muse_set_timing": (_i, [_vp, _i]),
    "muse_profile_begin": (_i, [_vp, _i]),
    "muse_profile_end": (_i, [_vp, C.POINTER(C.c_float), _i, C.POINTER(_i)]),
    "muse_profile_clock_hz": (_i, [_vp, C.POINTER(_d)]),
    "muse_sample_x_z": (_i, [_vp, _u64, _i64, _vp, _vp, _vp, _i]),
    "muse_logLike_and_grad_z": (_i, [_vp, _vp, _vp, _vp, C.POINTER(_d), _vp, _i]),
    "muse_grad_theta": (_i, [_vp, _vp, _vp, _vp, _vp, _i]),
    "muse_zhat_at_theta": (_i, [_vp, _vp, _vp, _vp, _d, _vp, _vp, _i]),
    "muse_map_and_score_batch": (_i, [_vp, _u64, _i64, _i64, _i, _vp, _d, _i, _vp, _vp]),
    "muse_map_and_score_batch_async": (_i, [_vp, _u64, _i64, _i64, _i, _vp, _d, _i, _i]),
    "muse_batch_wait": (_i, [_vp, _i, _vp, _vp]),
    "muse_map_and_score_multi_async": (_i, [_vp, _u64, _i64, _i64, _i, _i, _vp, _d, _i, _i]),
    "muse_map_and_score_multi_gather_async": (_i, [_vp, _u64, _i64, _i64, _i, _i, _vp, _d, _i, _i64, _i]),
    "muse_run": (_i, [_vp, _u64, _vp, _vp, C.POINTER(C.c_int32), _vp, _vp, _vp, _vp]),
    "muse_run_device": (_i, [_vp, _u64, _vp, _vp, C.POINTER(C.c_int32), _vp, _vp, _vp, _vp]),
    "muse_set_normals_cache": (_i, [_vp, _i]),
    "muse_set_constants": (_i, [_vp, _i, _vp, _i64, _i]),
    "muse_model_eval": (_i, [_vp, _d, _d, _d, _d, _d, _d, _i64, _vp]),
    "muse_model_has_second": (_i, []),
    "muse_run_sharded": (_i, [_vp, _u64, _vp, _vp, C.POINTER(C.c_int32), _vp, _vp, _vp, _vp]),
    "muse_get_zhat": (_i, [_vp, _i64, _i64, _vp, _i]),
    "muse_set_zhat": (_i, [_vp, _i64, _i64, _vp, _i]),
    "muse_fd_jacobian_batch": (_i, [_vp, _u64, _i64, _i64, _vp, _vp, _d, _i, _i64, _vp, _vp]),
    "muse_implicit_H_batch": (_i, [_vp, _u64, _i64, _i64, _vp, _d, _i, _vp, _vp]),
    "muse_fd_jacobian_columns": (_i, [_vp, _u64, _i64, _i64, _i64, _vp, _vp, _d, _i, _i64, _vp, _vp]),
    "muse_implicit_H_columns": (_i, [_vp, _u64, _i64, _i64, _i64, _vp, _d, _i, _vp, _vp]),
    "muse_fd_values_columns": (_i, [_vp, _u64, _i64, _i64, _i64, _vp, _i, _vp, _i, _d, _i, _i64, _vp, _vp]),
    "muse_comm_unique_id": (_i, [_vp]),
    "muse_comm_unique_id_ex": (_i, [_i, C.c_int64, _vp]),
    "muse_comm_transport": (_i, [_vp, C.POINTER(C.c_int)]),
    "muse_comm_ranks_seen": (_i, [_vp, C.POINTER(C.c_int)]),
    "muse_comm_init": (_i, [_vp, _i, _i, _vp]),
    "muse_comm_destroy": (_i, [_vp]),
    "muse_allgather_scores": (_i, [_vp, _vp, _i64, _vp]),
    "muse_allreduce_sum": (_i, [_vp, _vp, _i64]),
    "muse_map_and_score_batch_gather_async": (_i, [_vp, _u64, _i64, _i64, _i, _vp, _d, _i, _i64, _i]),
    "muse_batch_wait_gathered": (_i, [_vp, _i, _vp, _vp]),
    "muse_comm_board_status": (_i, [_vp, C.POINTER(_i), C.POINTER(_d)]),
    "muse_debug_flags": (_i, [_vp, _i]),
    "muse_debug_stamps": (_i, [_vp, _i64, _vp]),
}

_lib = None
_libs = {}   # path -> CDLL: libmuse_hip.so and the libraries built from users' model headers (models.ElementwiseModel)


def library_path():
    """The in-tree library; MUSE_HIP_LIB names an alternative build of the same sources (diagnostic builds)."""
    return os.environ.get("MUSE_HIP_LIB") or _build.LIB_PATH


def load_library(path=None):
    """dlopen libmuse_hip.so (built in-tree by build.build_extension) -- or, given `path`, the engine library of a user's
    model (build.build_model_library: the same ABI) -- and declare every signature."""
    global _lib
    if path is None:
        if _lib is not None:
            return _lib
        path = library_path()
    path = os.path.abspath(path)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise MuseError(-2, f"{path} is missing: run museinference.jl_amd.build.build_extension() "
                            "(there is no CPU fallback)")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _libs[path] = lib
    if path == os.path.abspath(library_path()):
        _lib = lib
    return lib


def check(rc, lib=None):
    """Raise MuseError for a negative status, with the message of the library the call went to."""
    if rc != 0:
        raise MuseError(rc, (lib or load_library()).muse_last_error().decode("utf-8", "replace"))


def f8(a, n=None):
    a = np.ascontiguousarray(np.atleast_1d(np.asarray(a, dtype=np.float64)))
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} doubles, got {a.size}")
    return a


def ptr(a):
    """void* of a numpy array, a raw integer device pointer, or an object with .data_ptr() (torch)."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(int(a))
