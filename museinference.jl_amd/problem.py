"""The problem interface (plugin/operator API) of the reference, src/interface.jl:4-186, and the
MI355X-backed implementation of it.

Name map (Julia identifiers that are not valid Python are transliterated; valid ones are aliased):
    sample_x_z              -> sample_x_z
    logLike_and_∇z_logLike  -> logLike_and_grad_z_logLike
    ∇θ_logLike              -> grad_theta_logLike
    ẑ_at_θ                  -> zhat_at_theta      (alias ẑ_at_θ)
    logPriorθ               -> logPrior_theta     (alias logPriorθ)
    transform_θ / inv_transform_θ / standardizeθ / ẑ_guess_from_truth
                            -> transform_theta / inv_transform_theta / standardize_theta /
                               zhat_guess_from_truth (aliases with θ/ẑ)
An `rng` is a SimRng(seed, sim): the stream of simulation `sim` under master seed `seed`
(split_rng, src/util.jl:87-92: a stream depends only on the master rng and the sim index).
"""
import ctypes as C
import warnings
from dataclasses import dataclass

import numpy as np

from . import _capi
from .priors import as_prior

UnTransformedθ = "untransformed"  # src/interface.jl:8-11
Transformedθ = "transformed"

MASTER_SIM = (1 << 62) - 1  # stream index standing for the un-split master rng (src/muse.jl:418)
DATA_SIM = (1 << 32) - 1    # stream used to draw synthetic "observed" data


@dataclass(frozen=True)
class SimRng:
    seed: int
    sim: int


def split_rng(rng, N):
    """split_rng(rng, N): N child streams, parent not advanced (src/util.jl:87-92)."""
    seed = rng.seed if isinstance(rng, SimRng) else int(rng)
    return [SimRng(seed, i) for i in range(N)]


def _seed_of(rng):
    return rng.seed if isinstance(rng, SimRng) else int(rng)


class AbstractMuseProblem:
    """abstract type AbstractMuseProblem (src/interface.jl:4) with the interface defaults."""

    x = None  # observed data, prob.x (src/muse.jl:170)

    # -- theta space transforms: identity defaults (src/interface.jl:20,28,134)
    def transform_theta(self, theta):
        return theta

    def inv_transform_theta(self, theta):
        return theta

    def standardize_theta(self, theta):
        return np.atleast_1d(np.asarray(theta, dtype=np.float64)).copy()

    # -- prior: zero default (src/interface.jl:120-121)
    def logPrior_theta(self, theta, theta_space=UnTransformedθ):
        return 0.0

    def grad_logPrior_theta(self, theta, theta_space=UnTransformedθ):
        return np.zeros_like(np.asarray(theta, dtype=np.float64))

    def hess_logPrior_theta(self, theta, theta_space=UnTransformedθ):
        n = np.asarray(theta).size
        return np.zeros((n, n))

    # -- per-simulation operators, to be implemented (src/interface.jl:41-99)
    def sample_x_z(self, rng, theta):
        raise NotImplementedError

    def logLike_and_grad_z_logLike(self, x, z, theta):
        raise NotImplementedError

    def grad_theta_logLike(self, x, z, theta, theta_space=UnTransformedθ):
        raise NotImplementedError

    def zhat_at_theta(self, x, z0, theta, grad_z_logLike_atol=1e-2):
        """The interface's DEFAULT (src/interface.jl:140-166): minimise -logLike over z from z0 with L-BFGS + HagerZhang (optim.py),
        g_tol = the tolerance, through the problem's own logLike_and_grad_z_logLike -- numpy vectors (solved on the host) or torch
        tensors (solved where they live).  HipMuseProblem overrides it with the HIP solver; a subclass that states its logLike and
        gradient gets this one, as a subclass of AbstractMuseProblem does in the reference."""
        import torch
        from . import _capi, optim
        as_np = not torch.is_tensor(z0)
        shape = np.shape(z0) if as_np else z0.shape
        z0t = torch.as_tensor(np.asarray(z0, dtype=np.float64)).reshape(-1) if as_np else z0.reshape(-1)

        def fg(z):
            f, g = self.logLike_and_grad_z_logLike(x, z.numpy().reshape(shape) if as_np else z.reshape(shape), theta)
            g = torch.as_tensor(np.asarray(g, dtype=np.float64)) if as_np else g
            return -float(f), -g.reshape(-1)
        z, info = optim.lbfgs(fg, z0t, grad_z_logLike_atol)
        rec = np.zeros((), dtype=_capi.INFO_DTYPE)
        for k in ("iterations", "f_calls", "status", "f_min", "gnorm"):
            rec[k] = info[k]
        return (z.numpy().reshape(shape) if as_np else z.reshape(shape)), rec

    def zhat_guess_from_truth(self, x, z, theta):
        """zero(z) (src/interface.jl:184-186)."""
        return np.zeros_like(z)

    # aliases with the reference's spelling where Python allows it
    ẑ_at_θ = property(lambda self: self.zhat_at_theta)
    logPriorθ = property(lambda self: self.logPrior_theta)
    transform_θ = property(lambda self: self.transform_theta)
    inv_transform_θ = property(lambda self: self.inv_transform_theta)
    standardizeθ = property(lambda self: self.standardize_theta)
    ẑ_guess_from_truth = property(lambda self: self.zhat_guess_from_truth)


def check_self_consistency(prob, theta, *, atol=1e-2, rng=None, has_volume_factor=True, step=1e-5):
    """check_self_consistency(prob, θ; fdm, atol, rng, has_volume_factor)   (src/interface.jl:209-230).

    Checks, for a freshly sampled (x, z):  inv_transform∘transform = id;  the prior in both θ spaces
    differs by the log-volume V(θ) = logdet J(θ), J = d transform/dθ;  and
    ∇θ logLike (untransformed) = Jᵀ ∇θ′ logLike (transformed) + ∇θ V.  J and ∇V by central differences,
    as the reference (it does not assume the transform is AD-able).  Returns a dict of the three
    residuals and raises AssertionError if one exceeds atol."""
    th = prob.standardize_theta(theta)
    n = th.size
    rng = SimRng(0, 0) if rng is None else rng
    x, z = prob.sample_x_z(rng, th)

    def jac(t):
        J = np.zeros((n, n))
        for j in range(n):
            e = np.zeros(n)
            e[j] = step
            J[:, j] = (np.atleast_1d(prob.transform_theta(t + e)) - np.atleast_1d(prob.transform_theta(t - e))) / (2 * step)
        return J

    def V(t):
        return np.linalg.slogdet(jac(t))[1] if has_volume_factor else 0.0

    gradV = np.zeros(n)
    if has_volume_factor:
        for j in range(n):
            e = np.zeros(n)
            e[j] = step
            gradV[j] = (V(th + e) - V(th - e)) / (2 * step)
    th_t = np.atleast_1d(prob.transform_theta(th))
    r1 = np.max(np.abs(np.atleast_1d(prob.inv_transform_theta(th_t)) - th))
    r2 = abs(prob.logPrior_theta(th, UnTransformedθ) - (prob.logPrior_theta(th_t, Transformedθ) + V(th)))
    g_u = np.atleast_1d(prob.grad_theta_logLike(x, z, th, UnTransformedθ))
    g_t = np.atleast_1d(prob.grad_theta_logLike(x, z, th_t, Transformedθ))
    r3 = np.max(np.abs(g_u - (jac(th).T @ g_t + gradV)))
    res = {"inv_transform": float(r1), "prior_volume": float(r2), "grad_chain_rule": float(r3)}
    for k, v in res.items():
        assert v <= atol, f"self-consistency check {k} failed: residual {v} > atol {atol}"
    return res


def check_optim_soln(info, where="MAP"):
    """_check_optim_soln (src/interface.jl:168-171): warn if not converged, log an error (no throw)
    if the minimum is not finite."""
    info = np.atleast_1d(info)
    bad = info["status"] >= _capi.STATUS_MAXITER
    if np.any(bad):
        warnings.warn(f"{where}: MAP solution did not converge within tolerance for {int(bad.sum())} "
                      "element(s), result could be erroneous. Try tweaking θ₀ or ∇z_logLike_atol "
                      "arguments to muse or fixing model.", RuntimeWarning, stacklevel=3)
    nf = ~np.isfinite(info["f_min"]) | (info["status"] == _capi.STATUS_NONFINITE)
    if np.any(nf):
        import logging
        logging.getLogger("museinference").error("%s: MAP solution failed with logjoint(MAP) non-finite "
                                                 "for %d element(s).", where, int(nf.sum()))


class HipMuseProblem(AbstractMuseProblem):
    """An AbstractMuseProblem whose operators run on one MI355X through libmuse_hip.so.

    Plays the role of SimpleMuseProblem (src/simple.jl:79-95) for the compiled-in models; on top of
    the per-simulation interface it exposes the two batched seams that replace the reference's
    pmap bodies: map_and_score_batch (src/muse.jl:169-176, :508-525) and fd_jacobian_batch
    (src/muse.jl:426-442).
    """

    supports_native_muse = True  # muse_() may hand the whole outer loop to muse_run (class attribute: wrappers
                                 # that forward attribute access to a HipMuseProblem do not inherit it)

    def __init__(self, x, model="funnel", ntheta=1, prior=None, device=0, N=None, constants=None):
        """constants: {"P": array of N doubles, ...} for a user model written with run-time constants
        (ElementwiseModel.from_source(..., runtime_constants=["P", ...]); include/muse_model.h, muse_const) -- what the
        reference's SimpleMuseProblem closures capture; set_constants replaces a vector later, without rebuilding anything."""
        from .models import ElementwiseModel
        if isinstance(model, ElementwiseModel):     # a user's model: its own engine library, model id MUSE_MODEL_USER
            self._lib = _capi.load_library(model.library())
            model_id, self.user_model, model = _capi.MODEL_USER, model, model.name
        else:
            self._lib = _capi.load_library()
            if model not in _capi.MODELS:
                raise ValueError(f"unknown model {model!r}; choose from {sorted(_capi.MODELS)} or pass an ElementwiseModel")
            model_id, self.user_model = _capi.MODELS[model], None
        if x is None and N is None:
            raise ValueError("give the observed data x (or N for a data-less problem)")
        self.x = None if x is None else _capi.f8(x)
        self.N = int(N) if x is None else int(self.x.size)
        self.model = model
        self.ntheta = int(ntheta)
        self.prior = as_prior(prior)
        self.device = int(device)
        ctx = C.c_void_p()
        self._check(self._lib.muse_ctx_create(model_id, self.N, self.ntheta, self.device, C.byref(ctx)))
        self._ctx = ctx
        if self.x is not None:
            self._check(self._lib.muse_set_data(self._ctx, _capi.ptr(self.x), _capi.MEM_HOST))
        for name, values in (constants or {}).items():
            self.set_constants(name, values)

    def set_constants(self, name, values):
        """Run-time constant vector `name` (or its index) of the user model: N finite doubles (muse_set_constants)."""
        names = getattr(self.user_model, "runtime_constants", None) or []
        k = names.index(name) if isinstance(name, str) else int(name)
        v = _capi.f8(values, self.N)
        self._check(self._lib.muse_set_constants(self._ctx, k, _capi.ptr(v), v.size, _capi.MEM_HOST))

    @property
    def has_second_derivatives(self):
        """Whether the implicit-differentiation get_H! accepts this problem's model: the built-in models, and a user-supplied
        header that defines MUSE_MODEL_SECOND (include/muse_model.h)."""
        return bool(self._lib.muse_model_has_second())

    def model_eval(self, iv, sd, x, z, n1, n2, i=0):
        """The functions of a user-supplied model's header at one element, evaluated on the host (muse_model_eval): a dict with
        grad (d(-logLike)/dz_i), term (A + iv B), B, ozz, ozx, bz, bx (muse_model_second) and z, x, dx_dsd of the draw at
        (sd, n1, n2).  What check_model_consistency differentiates numerically."""
        out = np.empty(12)
        self._check(self._lib.muse_model_eval(self._ctx, float(iv), float(sd), float(x), float(z), float(n1), float(n2), int(i),
                                              _capi.ptr(out)))
        if getattr(self.user_model, "pair", False):
            # a header of the two-parameter family (include/muse_model.h, MUSE_MODEL_PAIR): `iv` and `sd` were the block's parameters a, b
            return dict(zip(("grad", "term", "t0", "c0", "c1", "c2", "c3", "z", "x", "C", "t1"), out.tolist()))
        return dict(zip(("grad", "term", "B", "ozz", "ozx", "bz", "bx", "z", "x", "dx_dsd"), out.tolist()))

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.muse_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers
    def _check(self, rc):
        _capi.check(rc, self._lib)

    def _theta(self, theta):
        return _capi.f8(theta, self.ntheta)

    def set_placement(self, placement):
        """-1 auto, 0 streaming, 1 resident (bitwise-identical results; for tests/benchmarks)."""
        self._check(self._lib.muse_set_placement(self._ctx, int(placement)))

    def set_element_split(self, split):
        """Workgroups per map element: 0/1 = by N alone (default), 2/4/8/16 = that many workgroups share one
        element (for launches with fewer elements than compute units: the per-GPU share of a strongly scaled
        map, the axis choice of src/muse.jl:327-333).  Results depend on the split (summation tree), not on the
        batch or the GPU count."""
        self._check(self._lib.muse_set_element_split(self._ctx, int(split)))
        self.element_split = int(split)

    def set_concurrency(self, nlanes):
        """Result area r runs on lane r mod nlanes (a stream, scratch and ticket counter of its own): consecutive launches on
        different areas overlap instead of queueing behind each other's last workgroup.  Results are unchanged."""
        self._check(self._lib.muse_set_concurrency(self._ctx, int(nlanes)))

    def placement_info(self):
        """{'threads', 'workgroups_per_element', 'resident', 'direction_in_lds'} of this problem's batched maps."""
        v = [C.c_int() for _ in range(4)]
        self._check(self._lib.muse_placement_info(self._ctx, *[C.byref(x) for x in v]))
        return dict(threads=v[0].value, workgroups_per_element=v[1].value, resident=bool(v[2].value),
                    direction_in_lds=bool(v[3].value))

    def set_stream(self, hip_stream):
        self._check(self._lib.muse_set_stream(self._ctx, _capi.ptr(hip_stream) if hip_stream else None))

    def synchronize(self):
        self._check(self._lib.muse_synchronize(self._ctx))

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(self._lib.muse_last_kernel_ms(self._ctx, C.byref(ms)))
        return ms.value

    def set_timing(self, enabled):
        self._check(self._lib.muse_set_timing(self._ctx, int(bool(enabled))))

    def profile_begin(self, max_launches=1024):
        self._check(self._lib.muse_profile_begin(self._ctx, int(max_launches)))
        self._prof_cap = int(max_launches)

    def profile_end(self):
        """Per-launch solver kernel durations (ms) since profile_begin (HIP events on the launch stream)."""
        buf = (C.c_float * self._prof_cap)()
        n = C.c_int()
        self._check(self._lib.muse_profile_end(self._ctx, buf, self._prof_cap, C.byref(n)))
        return np.array(buf[: min(n.value, self._prof_cap)], dtype=np.float64)

    def profile_clock_hz(self):
        """Shader clock during the last launch profiled between profile_begin and profile_end (in-kernel counters)."""
        hz = C.c_double()
        self._check(self._lib.muse_profile_clock_hz(self._ctx, C.byref(hz)))
        return hz.value

    # -- prior (SimpleMuseProblem forwards to the user's function, src/simple.jl:93)
    def logPrior_theta(self, theta, theta_space=UnTransformedθ):
        return self.prior.logpdf(np.asarray(theta, dtype=np.float64))

    def grad_logPrior_theta(self, theta, theta_space=UnTransformedθ):
        return np.atleast_1d(self.prior.grad(np.asarray(theta, dtype=np.float64)))

    def hess_logPrior_theta(self, theta, theta_space=UnTransformedθ):
        return np.atleast_2d(self.prior.hess(np.asarray(theta, dtype=np.float64)))

    # -- per-simulation operators
    def sample_x_z(self, rng, theta):
        """sample_x_z(prob, rng, θ) -> (x, z)   [src/interface.jl:92-99]"""
        x = np.empty(self.N)
        z = np.empty(self.N)
        th = self._theta(theta)
        self._check(self._lib.muse_sample_x_z(self._ctx, rng.seed, rng.sim, _capi.ptr(th), _capi.ptr(x),
                                              _capi.ptr(z), _capi.MEM_HOST))
        return x, z

    def logLike_and_grad_z_logLike(self, x, z, theta):
        """logLike_and_∇z_logLike(prob, x, z, θ) -> (logLike, ∇z logLike)   [src/interface.jl:68-83]"""
        x = _capi.f8(x, self.N)
        z = _capi.f8(z, self.N)
        th = self._theta(theta)
        g = np.empty(self.N)
        f = C.c_double()
        self._check(self._lib.muse_logLike_and_grad_z(self._ctx, _capi.ptr(x), _capi.ptr(z), _capi.ptr(th),
                                                      C.byref(f), _capi.ptr(g), _capi.MEM_HOST))
        return f.value, g

    def grad_theta_logLike(self, x, z, theta, theta_space=UnTransformedθ):
        """∇θ_logLike(prob, x, z, θ[, θ_space])   [src/interface.jl:41-58]"""
        x = _capi.f8(x, self.N)
        z = _capi.f8(z, self.N)
        th = self._theta(theta)
        g = np.empty(self.ntheta)
        self._check(self._lib.muse_grad_theta(self._ctx, _capi.ptr(x), _capi.ptr(z), _capi.ptr(th), _capi.ptr(g),
                                              _capi.MEM_HOST))
        return g

    def zhat_at_theta(self, x, z0, theta, grad_z_logLike_atol=1e-2):
        """ẑ_at_θ(prob, x, z₀, θ; ∇z_logLike_atol) -> (ẑ, info)   [src/interface.jl:141-171]"""
        x = _capi.f8(x, self.N)
        z0 = _capi.f8(z0, self.N)
        th = self._theta(theta)
        z = np.empty(self.N)
        info = np.zeros(1, dtype=_capi.INFO_DTYPE)
        self._check(self._lib.muse_zhat_at_theta(self._ctx, _capi.ptr(x), _capi.ptr(z0), _capi.ptr(th),
                                                 float(grad_z_logLike_atol), _capi.ptr(z), _capi.ptr(info),
                                                 _capi.MEM_HOST))
        check_optim_soln(info, "ẑ_at_θ")
        return z, info[0]

    # -- batched seams
    def map_and_score_batch(self, rng, sim_begin, sim_end, theta, *, include_data=False, atol=1e-2,
                            z0_mode=_capi.Z0_ZERO):
        """All elements of one muse!/get_J! map in one launch.  Returns (g [n, nθ], info [n]);
        element order: [data], sim_begin .. sim_end-1.  ẑ of element e stays resident at slot e."""
        th = self._theta(theta)
        n = (sim_end - sim_begin) + (1 if include_data else 0)
        g = np.empty((n, self.ntheta))
        info = np.zeros(n, dtype=_capi.INFO_DTYPE)
        self._check(self._lib.muse_map_and_score_batch(self._ctx, _seed_of(rng), sim_begin, sim_end,
                                                       int(bool(include_data)), _capi.ptr(th), float(atol),
                                                       int(z0_mode), _capi.ptr(g), _capi.ptr(info)))
        return g, info

    def map_and_score_batch_async(self, rng, sim_begin, sim_end, theta, *, include_data=False, atol=1e-2,
                                  z0_mode=_capi.Z0_ZERO, result_area=0):
        th = self._theta(theta)
        self._check(self._lib.muse_map_and_score_batch_async(self._ctx, _seed_of(rng), sim_begin, sim_end,
                                                             int(bool(include_data)), _capi.ptr(th), float(atol),
                                                             int(z0_mode), int(result_area)))
        return (sim_end - sim_begin) + (1 if include_data else 0)

    def map_and_score_multi_async(self, rng, sim_begin, sim_end, thetas, *, include_data=False, atol=1e-2,
                                  z0_mode=_capi.Z0_ZERO, result_area=0):
        """Several independent maps over the same elements in ONE launch, map m at thetas[m] (muse_map_and_score_multi_async
        of the C ABI: a launch with fewer elements than compute units leaves the GPU idle for one problem's latency;
        several such maps resident at once fill it).  Returns the total row count nmaps * n for batch_wait."""
        th = np.ascontiguousarray(np.asarray(thetas, dtype=np.float64).reshape(-1, self.ntheta))
        if not 1 <= th.shape[0] <= _capi.MAX_MAPS:
            raise ValueError(f"1 <= nmaps <= {_capi.MAX_MAPS}")
        self._check(self._lib.muse_map_and_score_multi_async(self._ctx, _seed_of(rng), sim_begin, sim_end,
                                                             int(bool(include_data)), th.shape[0], _capi.ptr(th), float(atol),
                                                             int(z0_mode), int(result_area)))
        return th.shape[0] * ((sim_end - sim_begin) + (1 if include_data else 0))

    def batch_wait(self, n, result_area=0, out=None):
        """out: (g [n, nθ] float64, info [n] INFO_DTYPE), C-contiguous arrays to fill instead of fresh ones (a pipelined
        host loop that waits every ~20 us reuses one pair per result area)."""
        if out is None:
            g = np.empty((n, self.ntheta))
            info = np.zeros(n, dtype=_capi.INFO_DTYPE)
        else:
            g, info = out
            if g.shape != (n, self.ntheta) or info.shape != (n,) or g.dtype != np.float64 or info.dtype != _capi.INFO_DTYPE \
                    or not (g.flags.c_contiguous and info.flags.c_contiguous):
                raise ValueError("out must be (float64 [n, ntheta], INFO_DTYPE [n]), C-contiguous")
        self._check(self._lib.muse_batch_wait(self._ctx, int(result_area), _capi.ptr(g), _capi.ptr(info)))
        return g, info

    def native_prior(self):
        """(kind, mean[nθ], sigma[nθ]) if the prior is one muse_run evaluates itself, else None."""
        from .priors import FlatPrior, GaussianPrior
        if self.ntheta > _capi.MAX_THETA:        # (the big tier of muse_hip.h: the loop over the batched maps runs in muse.py)
            return None
        if type(self.prior) is FlatPrior:
            return 0, np.zeros(self.ntheta), np.ones(self.ntheta)
        if type(self.prior) is GaussianPrior:
            return (1, np.broadcast_to(np.asarray(self.prior.mean, dtype=np.float64), (self.ntheta,)).copy(),
                    np.broadcast_to(np.asarray(self.prior.sigma, dtype=np.float64), (self.ntheta,)).copy())
        return None

    def run_muse(self, rng, theta0, *, nsims, maxsteps, theta_rtol, atol, alpha, z0_warm=False, device_loop=None):
        """The muse! outer loop in the library's native code (muse_run / muse_run_device, include/muse_hip.h): returns
        (n, theta, hist [n, W], g_sims [n, nsims, nθ], info [n, nsims+1]).  device_loop=True: ONE launch runs every
        iteration -- map, exchange of the scores between the (all resident) workgroups, step, next map -- and nothing leaves
        the GPU in between (the default: 42 against 48 us per steady iteration at N = 10^4 x 512 sims, and no host in the loop to be
        slowed by whatever else the process does); False: one launch per iteration, the algebra on the host.  The same results
        bit for bit either way; placements other than the resident ones (an element split, N > 10 000) and five to eight theta
        components with several elements per workgroup run the host loop whatever is asked; so does a context whose loop kernel
        once failed to keep its workgroups resident (a shared GPU)."""
        native = self.native_prior()
        if native is None:
            raise _capi.MuseError(-1, "the native muse! loops take a flat or Gaussian prior and ntheta <= MUSE_MAX_THETA "
                                                     "(muse() runs its loop over the batched maps otherwise)")
        kind, mean, sigma = native
        o = _capi.RunOptions()
        o.nsims, o.maxsteps, o.theta_rtol, o.atol, o.alpha = int(nsims), int(maxsteps), float(theta_rtol), float(atol), float(alpha)
        o.prior_kind, o.z0_warm = int(kind), int(bool(z0_warm))
        for k in range(self.ntheta):
            o.prior_mean[k], o.prior_sigma[k] = float(mean[k]), float(sigma[k])
        th0 = self._theta(theta0)
        W = _capi.run_hist_width(self.ntheta)
        # (not zero-filled: the library writes every row it reports, the others are cut off below -- at 30 x 513 elements the
        #  fill of the info block alone was 40 us of a 1.5 ms call)
        hist = np.empty((maxsteps, W))
        gs = np.empty((maxsteps, nsims, self.ntheta))
        info = np.empty((maxsteps, nsims + 1), dtype=_capi.INFO_DTYPE)
        theta = np.zeros(self.ntheta)
        n = C.c_int32()
        auto = device_loop is None
        device_loop = True if auto else bool(device_loop)   # default: the device-resident loop (measured faster)
        fn = self._lib.muse_run_device if device_loop else self._lib.muse_run
        args = (self._ctx, _seed_of(rng), _capi.ptr(th0), C.byref(o), C.byref(n), _capi.ptr(theta), _capi.ptr(hist), _capi.ptr(gs),
                _capi.ptr(info))
        try:
            self._check(fn(*args))
        except _capi.MuseError as e:
            # The loop kernel needs all of its workgroups resident at once; on a GPU that something else is using they may
            # not be, its bounded waits expire and the run is reported as failed.  When the caller left the choice of loop to
            # the library -- and the run did not start from the resident MAPs, which the aborted attempt has touched -- the
            # host loop takes over: the same bits as the device loop would have given.
            if not (auto and not z0_warm and "not all resident" in str(e)):
                raise
            warnings.warn("museinference: the device-resident muse! loop could not keep its workgroups resident (is the GPU shared?); "
                          "running the host loop instead", RuntimeWarning)
            self._check(self._lib.muse_run(*args))
        return n.value, theta, hist[: n.value], gs[: n.value], info[: n.value]

    def set_normals_cache(self, enabled):
        """Plain maps over a simulation range the context has just been asked for store (second time) and then load (from
        the third time on) the streams' standard normals instead of generating them again -- muse_set_normals_cache;
        False switches that off (a benchmark that repeats one map to time the generator)."""
        self._check(self._lib.muse_set_normals_cache(self._ctx, int(bool(enabled))))

    def run_muse_sharded(self, rng, theta0, *, nsims, maxsteps, theta_rtol, atol, alpha, z0_warm=False):
        """This rank's part of the muse! loop over the ranks of the context's communicator (muse_run_sharded; comm_init
        first): (n, theta, hist, g_sims) as run_muse -- the same on every rank -- and THIS rank's solver infos
        [n, count of its elements] (rank 0: the data element first).  With the shared-memory transport and a resident placement
        the loop is ONE persistent launch per rank (scores exchanged through boards in device memory that the ranks map into each
        other by hipIpc, or through one board in pinned host memory); the host-driven loop otherwise -- the same bits."""
        native = self.native_prior()
        if native is None:
            raise _capi.MuseError(-1, "the native muse! loops take a flat or Gaussian prior and ntheta <= MUSE_MAX_THETA "
                                                     "(muse() runs its loop over the batched maps otherwise)")
        kind, mean, sigma = native
        o = _capi.RunOptions()
        o.nsims, o.maxsteps, o.theta_rtol, o.atol, o.alpha = int(nsims), int(maxsteps), float(theta_rtol), float(atol), float(alpha)
        o.prior_kind, o.z0_warm = int(kind), int(bool(z0_warm))
        for k in range(self.ntheta):
            o.prior_mean[k], o.prior_sigma[k] = float(mean[k]), float(sigma[k])
        th0 = self._theta(theta0)
        world, rank = self._nranks, self._rank
        base, extra = divmod(int(nsims), world)
        nloc = base + (1 if rank < extra else 0) + (1 if rank == 0 else 0)
        W = _capi.run_hist_width(self.ntheta)
        hist = np.zeros((maxsteps, W))
        gs = np.zeros((maxsteps, nsims, self.ntheta))
        info = np.zeros((maxsteps, nloc), dtype=_capi.INFO_DTYPE)
        theta = np.zeros(self.ntheta)
        n = C.c_int32()
        self._check(self._lib.muse_run_sharded(self._ctx, _seed_of(rng), _capi.ptr(th0), C.byref(o), C.byref(n), _capi.ptr(theta),
                                               _capi.ptr(hist), _capi.ptr(gs), _capi.ptr(info)))
        return n.value, theta, hist[: n.value], gs[: n.value], info[: n.value]

    def get_zhat(self, slot_begin, slot_end):
        out = np.empty((slot_end - slot_begin, self.N))
        self._check(self._lib.muse_get_zhat(self._ctx, slot_begin, slot_end, _capi.ptr(out), _capi.MEM_HOST))
        return out

    def set_zhat(self, slot_begin, zs):
        zs = np.ascontiguousarray(np.atleast_2d(np.asarray(zs, dtype=np.float64)))
        if zs.shape[1] != self.N:
            raise ValueError("zhat rows must have N columns")
        self._check(self._lib.muse_set_zhat(self._ctx, slot_begin, slot_begin + zs.shape[0], _capi.ptr(zs),
                                            _capi.MEM_HOST))

    def fd_jacobian_batch(self, rng, sim_begin, sim_end, theta0, step, *, atol=1e-2, fid_mode=0,
                          fid_sim=MASTER_SIM):
        """get_H! finite-difference Jacobians for sims [sim_begin, sim_end): returns
        (Hs [nsims, nθ, nθ], info [nsims, nθ, 2])."""
        th = self._theta(theta0)
        st = _capi.f8(step, self.ntheta)
        ns = sim_end - sim_begin
        Hs = np.empty((ns, self.ntheta, self.ntheta))
        info = np.zeros((ns, self.ntheta, 2), dtype=_capi.INFO_DTYPE)
        self._check(self._lib.muse_fd_jacobian_batch(self._ctx, _seed_of(rng), sim_begin, sim_end, _capi.ptr(th),
                                                     _capi.ptr(st), float(atol), int(fid_mode), int(fid_sim),
                                                     _capi.ptr(Hs), _capi.ptr(info)))
        return Hs, info

    def implicit_H_batch(self, rng, sim_begin, sim_end, theta0, *, atol=1e-1, cg_maxiter=100):
        """get_H! implicit-differentiation branch for sims [sim_begin, sim_end): (Hs [nsims, nθ, nθ],
        cg iteration counts [nsims, nθ])   [src/muse.jl:335-405]"""
        th = self._theta(theta0)
        ns = sim_end - sim_begin
        Hs = np.empty((ns, self.ntheta, self.ntheta))
        its = np.zeros((ns, self.ntheta), dtype=np.int32)
        self._check(self._lib.muse_implicit_H_batch(self._ctx, _seed_of(rng), sim_begin, sim_end, _capi.ptr(th),
                                                    float(atol), int(cg_maxiter), _capi.ptr(Hs), _capi.ptr(its)))
        return Hs, its

    def fd_jacobian_columns(self, rng, sim_begin, col_begin, col_end, theta0, step, *, atol=1e-2, fid_mode=0,
                            fid_sim=MASTER_SIM):
        """Columns [col_begin, col_end) of the list (sim_begin, column 0), (sim_begin, column 1), ...: the reference's
        other parallel axis for get_H! (Jacobian columns, src/muse.jl:327-333).  Returns (cols [n, nθ] with
        cols[e][i] = d g_i / d θ_(e mod nθ), info [n, 2])."""
        th = self._theta(theta0)
        st = _capi.f8(step, self.ntheta)
        n = col_end - col_begin
        cols = np.empty((n, self.ntheta))
        info = np.zeros((n, 2), dtype=_capi.INFO_DTYPE)
        self._check(self._lib.muse_fd_jacobian_columns(self._ctx, _seed_of(rng), sim_begin, col_begin, col_end, _capi.ptr(th),
                                                       _capi.ptr(st), float(atol), int(fid_mode), int(fid_sim),
                                                       _capi.ptr(cols), _capi.ptr(info)))
        return cols, info

    def fd_values_columns(self, rng, sim_begin, col_begin, col_end, theta0, offsets, *, per_unit=False, atol=1e-2, fid_mode=0,
                          fid_sim=MASTER_SIM):
        """The raw values of get_H!'s finite-difference map (muse_fd_values_columns): for units (sim, column) [col_begin,
        col_end) of the list and G grid points each, the score at theta0 of the simulation drawn at theta0 + offset e_j.
        offsets [nθ, G] (row j for column j, shared by the sims) or, per_unit, [n, G] (one row per unit).  Returns
        (F [n, G, nθ], info [n, G])."""
        th = self._theta(theta0)
        off = np.ascontiguousarray(np.asarray(offsets, dtype=np.float64))
        n = col_end - col_begin
        if n < 0:
            raise ValueError("col_end < col_begin")
        if off.ndim != 2 or off.shape[0] != (n if per_unit else self.ntheta):
            raise ValueError("offsets must be [ntheta, G], or [n_units, G] with per_unit")
        G = off.shape[1]
        F = np.empty((n, G, self.ntheta))
        info = np.zeros((n, G), dtype=_capi.INFO_DTYPE)
        self._check(self._lib.muse_fd_values_columns(self._ctx, _seed_of(rng), sim_begin, col_begin, col_end, _capi.ptr(th), G,
                                                     _capi.ptr(off), int(per_unit), float(atol), int(fid_mode), int(fid_sim),
                                                     _capi.ptr(F), _capi.ptr(info)))
        return F, info

    def implicit_H_columns(self, rng, sim_begin, col_begin, col_end, theta0, *, atol=1e-1, cg_maxiter=100):
        """The same column range for the implicit-differentiation H: (cols [n, nθ], cg iteration counts [n])."""
        th = self._theta(theta0)
        n = col_end - col_begin
        cols = np.empty((n, self.ntheta))
        its = np.zeros(n, dtype=np.int32)
        self._check(self._lib.muse_implicit_H_columns(self._ctx, _seed_of(rng), sim_begin, col_begin, col_end, _capi.ptr(th),
                                                      float(atol), int(cg_maxiter), _capi.ptr(cols), _capi.ptr(its)))
        return cols, its

    # -- exchange between ranks (C1-C3 of SURVEY.md §2) through the C ABI, for hosts without torch.distributed
    TRANSPORTS = {"rccl": 0, "shm": 1}

    @staticmethod
    def comm_unique_id(transport="rccl", block_doubles=0):
        """Created on rank 0 and passed (by any means) to every rank's comm_init.  transport "rccl": collectives
        over xGMI; "shm": the ranks of one node exchange their blocks through a shared-memory segment."""
        buf = (C.c_char * _capi.UNIQUE_ID_BYTES)()
        _capi.check(_capi.load_library().muse_comm_unique_id_ex(HipMuseProblem.TRANSPORTS[transport], int(block_doubles), buf))
        return bytes(buf)

    def comm_init(self, nranks, rank, unique_id):
        if len(unique_id) != _capi.UNIQUE_ID_BYTES:
            raise ValueError("unique_id must be the bytes returned by comm_unique_id")
        buf = (C.c_char * _capi.UNIQUE_ID_BYTES).from_buffer_copy(unique_id)  # (binary: not a C string)
        self._check(self._lib.muse_comm_init(self._ctx, int(nranks), int(rank), buf))
        self._nranks, self._rank = int(nranks), int(rank)

    def comm_destroy(self):
        """Tear the communicator down (a context may then be given another one, e.g. the other transport)."""
        self._check(self._lib.muse_comm_destroy(self._ctx))
        self._nranks = None

    def comm_ranks_seen(self):
        """Ranks the communicator itself counts (ncclCommCount / processes attached to the shared segment)."""
        n = C.c_int(-1)
        self._check(self._lib.muse_comm_ranks_seen(self._ctx, C.byref(n)))
        return n.value

    BOARDS = {0: "none", 1: "host", 2: "device"}
    # host-side bits of debug_flags (csrc/switches.hpp)
    DEBUG_HOST_BOARD, DEBUG_SHARDED_HOST_LOOP, DEBUG_LOOP_OVERSUBSCRIBE, DEBUG_RUN_TIMING = 1 << 16, 1 << 17, 1 << 18, 1 << 19

    def debug_flags(self, flags):
        """Diagnostic switches of this live context (include/muse_hip.h, "diagnostics": no bit changes a result)."""
        self._check(self._lib.muse_debug_flags(self._ctx, int(flags)))

    def comm_board_status(self):
        """The score boards of the sharded muse! loop (muse_comm_board_status): COLLECTIVE over a shared-memory communicator the first
        time it (or run_muse_sharded) is called -- the boards are mapped and each kind is proved by a millisecond hand-shake between
        the ranks' GPUs.  Returns {"board": "device" | "host" | "none" (what the persistent loop will use), "device_handshake" /
        "host_handshake": 1 ok, 0 failed, -1 not tried, "device_seen" / "host_seen": bit mask of the ranks this rank saw,
        "device_wait_us" / "host_wait_us", "last_loop": what the last run_muse_sharded call ran}."""
        st = (C.c_int * 6)()
        w = (C.c_double * 2)()
        self._check(self._lib.muse_comm_board_status(self._ctx, st, w))
        return {"board": self.BOARDS[st[0]], "device_handshake": st[1], "host_handshake": st[2], "device_seen": st[3], "host_seen": st[4],
                "last_loop": self.BOARDS[st[5]], "device_wait_us": w[0], "host_wait_us": w[1]}

    def comm_transport(self):
        t = C.c_int(-1)
        self._check(self._lib.muse_comm_transport(self._ctx, C.byref(t)))
        return {v: k for k, v in self.TRANSPORTS.items()}[t.value]

    def allgather_scores(self, send):
        send = _capi.f8(send)
        recv = np.empty((self._nranks, send.size))
        self._check(self._lib.muse_allgather_scores(self._ctx, _capi.ptr(send), send.size, _capi.ptr(recv)))
        return recv

    def map_and_score_batch_gather_async(self, rng, sim_begin, sim_end, theta, rows_per_rank, *, include_data=False,
                                         atol=1e-2, z0_mode=_capi.Z0_ZERO, result_area=0):
        """This rank's block of a sharded map; the score blocks of all ranks are all-gathered (RCCL: on the device,
        on the communicator's own stream; shm: between the hosts, inside batch_wait_gathered).  Returns this rank's
        element count."""
        th = self._theta(theta)
        self._check(self._lib.muse_map_and_score_batch_gather_async(
            self._ctx, _seed_of(rng), sim_begin, sim_end, int(bool(include_data)), _capi.ptr(th), float(atol),
            int(z0_mode), int(rows_per_rank), int(result_area)))
        return (sim_end - sim_begin) + (1 if include_data else 0)

    def map_and_score_multi_gather_async(self, rng, sim_begin, sim_end, thetas, rows_per_rank, *, include_data=False,
                                         atol=1e-2, z0_mode=_capi.Z0_ZERO, result_area=0):
        """The sharded form of map_and_score_multi_async: this rank's block of nmaps maps in one launch and ONE exchange
        for all of them; batch_wait_gathered(nmaps * n, nmaps * rows_per_rank, area) then returns
        g_all [nranks, nmaps * rows_per_rank, nθ] (map m of rank q: rows m*rows_per_rank ...) and info [nmaps * n]."""
        th = np.ascontiguousarray(np.asarray(thetas, dtype=np.float64).reshape(-1, self.ntheta))
        self._check(self._lib.muse_map_and_score_multi_gather_async(
            self._ctx, _seed_of(rng), sim_begin, sim_end, int(bool(include_data)), th.shape[0], _capi.ptr(th), float(atol),
            int(z0_mode), int(rows_per_rank), int(result_area)))
        return th.shape[0] * ((sim_end - sim_begin) + (1 if include_data else 0))

    def batch_wait_gathered(self, n, rows_per_rank, result_area=0, out=None):
        """(g_all [nranks, rows_per_rank, nθ], this rank's info [n]) of the gather enqueued on result_area; out: arrays
        of those shapes to fill instead of fresh ones."""
        if out is None:
            g = np.empty((self._nranks, int(rows_per_rank), self.ntheta))
            info = np.zeros(n, dtype=_capi.INFO_DTYPE)
        else:
            g, info = out
            if g.shape != (self._nranks, int(rows_per_rank), self.ntheta) or info.shape != (n,) or g.dtype != np.float64 \
                    or info.dtype != _capi.INFO_DTYPE or not (g.flags.c_contiguous and info.flags.c_contiguous):
                raise ValueError("out must be (float64 [nranks, rows_per_rank, ntheta], INFO_DTYPE [n]), C-contiguous")
        self._check(self._lib.muse_batch_wait_gathered(self._ctx, int(result_area), _capi.ptr(g), _capi.ptr(info)))
        return g, info

    def allreduce_sum(self, buf):
        buf = _capi.f8(buf).copy()
        self._check(self._lib.muse_allreduce_sum(self._ctx, _capi.ptr(buf), buf.size))
        return buf


class PositiveThetaProblem(AbstractMuseProblem):
    """A bounded-θ front-end: the user's parameters are the VARIANCES v_k = e^{θ_k} > 0 of the engine's
    models; the transformed space (domain (-inf, inf)) is the engine's own θ = log v.  Exercises the
    Transformedθ / UnTransformedθ machinery of the reference (src/interface.jl:8-28,41-58,107-121; the
    Turing adapter is its precedent, src/turing.jl:171-186): muse! iterates in the transformed space, the
    map bodies receive untransformed θ, the result is reported untransformed.

    `prior` is the prior density over the variances v (an object with logpdf/grad/hess); in the transformed
    space it picks up the log-volume factor: logPrior′(θ′) = logPrior(e^{θ′}) + Σ θ′.
    """

    def __init__(self, base, prior=None):
        from .priors import as_prior
        self.base = base
        self.x = base.x
        self.prior_v = as_prior(prior)

    # Attributes of the wrapped problem that mean the same thing in the variance parametrisation.  Anything that
    # takes or returns theta-space quantities (the batched seams, the per-simulation operators) is defined
    # explicitly below with its chain rule -- a blanket forward would hand variances to an engine that expects
    # log-variances.
    _FORWARDED = ("N", "ntheta", "model", "device", "get_zhat", "set_zhat", "close", "synchronize", "set_timing",
                  "last_kernel_ms", "set_placement", "set_element_split")

    def __getattr__(self, name):
        if name in PositiveThetaProblem._FORWARDED:
            return getattr(self.base, name)
        raise AttributeError(f"{type(self).__name__} has no attribute {name!r}")

    def standardize_theta(self, v):
        return np.atleast_1d(np.asarray(v, dtype=np.float64)).copy()

    def transform_theta(self, v):
        return np.log(np.asarray(v, dtype=np.float64))

    def inv_transform_theta(self, t):
        return np.exp(np.asarray(t, dtype=np.float64))

    def logPrior_theta(self, theta, theta_space=UnTransformedθ):
        if theta_space == Transformedθ:
            t = np.atleast_1d(np.asarray(theta, dtype=np.float64))
            return self.prior_v.logpdf(np.exp(t)) + float(np.sum(t))
        return self.prior_v.logpdf(np.atleast_1d(np.asarray(theta, dtype=np.float64)))

    def grad_logPrior_theta(self, theta, theta_space=UnTransformedθ):
        if theta_space == Transformedθ:
            t = np.atleast_1d(np.asarray(theta, dtype=np.float64))
            v = np.exp(t)
            return np.atleast_1d(self.prior_v.grad(v)) * v + 1.0
        return np.atleast_1d(self.prior_v.grad(np.atleast_1d(np.asarray(theta, dtype=np.float64))))

    def hess_logPrior_theta(self, theta, theta_space=UnTransformedθ):
        if theta_space == Transformedθ:
            t = np.atleast_1d(np.asarray(theta, dtype=np.float64))
            v = np.exp(t)
            H = np.atleast_2d(self.prior_v.hess(v))
            return np.outer(v, v) * H + np.diag(np.atleast_1d(self.prior_v.grad(v)) * v)
        return np.atleast_2d(self.prior_v.hess(np.atleast_1d(np.asarray(theta, dtype=np.float64))))

    # per-simulation operators: θ arrives untransformed (variances) except where θ_space says otherwise
    def sample_x_z(self, rng, v):
        return self.base.sample_x_z(rng, np.log(np.atleast_1d(v)))

    def logLike_and_grad_z_logLike(self, x, z, v):
        return self.base.logLike_and_grad_z_logLike(x, z, np.log(np.atleast_1d(v)))

    def zhat_at_theta(self, x, z0, v, grad_z_logLike_atol=1e-2):
        return self.base.zhat_at_theta(x, z0, np.log(np.atleast_1d(v)), grad_z_logLike_atol)

    def grad_theta_logLike(self, x, z, theta, theta_space=UnTransformedθ):
        th = np.atleast_1d(np.asarray(theta, dtype=np.float64))
        if theta_space == Transformedθ:
            # d/dθ′ with θ′ = log v, plus the gradient of the log-volume term (+Σθ′) that the transformed
            # density carries in the reference's convention (the logjoint of src/turing.jl:188-192); it is
            # the same constant for the data and every simulation, so it cancels in the MUSE gradient
            return self.base.grad_theta_logLike(x, z, th) + 1.0
        return self.base.grad_theta_logLike(x, z, np.log(th)) / th       # d/dv = (d/dθ′) / v

    # batched seams: the engine works in θ′ = log v
    def map_and_score_batch(self, rng, sim_begin, sim_end, v, **kw):
        return self.base.map_and_score_batch(rng, sim_begin, sim_end, np.log(np.atleast_1d(v)), **kw)

    def scores_in_both_spaces(self, g_engine, v, theta_t):
        """The engine's score is d logLike/dθ′; untransformed g = that / v, transformed g′ = that + 1."""
        return g_engine / np.atleast_1d(v)[None, :], g_engine + 1.0

    def fd_jacobian_batch(self, rng, sim_begin, sim_end, v0, step, **kw):
        """Finite differences in the untransformed space through the engine's θ′-space FD is not the same
        stencil; fall back to the element-by-element path of get_H_ (which calls the per-sim operators)."""
        raise NotImplementedError

    def implicit_H_batch(self, rng, sim_begin, sim_end, v0, **kw):
        """The implicit-differentiation H (src/muse.jl:335-405) in the variance parametrisation: the engine
        returns H′[i][j] = d g′_i / d θ′_j at θ′ = log v0 (the derivative acts on the sampling θ only, the score
        is taken at the fixed θ0); with g_i = g′_i / v0_i and d/dv_j = (1/v0_j) d/dθ′_j,  H = H′ / (v0 v0ᵀ)."""
        if not hasattr(self.base, "implicit_H_batch"):
            raise NotImplementedError("the wrapped problem has no implicit_H_batch seam")
        v0 = np.atleast_1d(np.asarray(v0, dtype=np.float64))
        Hs, its = self.base.implicit_H_batch(rng, sim_begin, sim_end, np.log(v0), **kw)
        return Hs / np.outer(v0, v0)[None, :, :], its
