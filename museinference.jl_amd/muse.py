"""Host drivers: MuseResult, muse / muse_ (muse!), get_J_ (get_J!), get_H_ (get_H!), finalize_result_.

A restatement of the post-map algebra of the reference (src/muse.jl:29-59, 107-250, 296-333,
407-450, 484-549): everything *inside* the reference's pmap bodies runs on the GPU through the
problem's batched seams (HipMuseProblem.map_and_score_batch / fd_jacobian_batch); everything after
the map is O(nsims * nθ²) numpy algebra, as in the reference.  Julia's in-place `f!` is spelled `f_`.

A problem that does not provide the batched seams (any other AbstractMuseProblem subclass) is driven
element by element through sample_x_z / zhat_at_theta / grad_theta_logLike, i.e. the reference's
LocalWorkerPool path (src/util.jl:74-76).
"""
import math
import pickle
import time
import warnings
from dataclasses import dataclass, field

import numpy as np

from . import _capi
from .fdm import as_fdm
from .problem import MASTER_SIM, SimRng, UnTransformedθ, Transformedθ, split_rng


@dataclass
class Normal:
    """Normal / MvNormal built from θ and Σ (src/muse.jl:542-546)."""
    mean: np.ndarray
    cov: np.ndarray

    @property
    def mu(self):
        return self.mean if self.mean.size > 1 else float(self.mean[0])

    @property
    def sigma(self):
        s = np.sqrt(np.diag(self.cov))
        return s if s.size > 1 else float(s[0])

    μ = mu
    σ = sigma


@dataclass
class MuseResult:
    """Result and resume state of a MUSE run (src/muse.jl:29-42).  `rng` is the master seed."""
    theta: np.ndarray = None
    H: np.ndarray = None
    J: np.ndarray = None
    Sigma_inv: np.ndarray = None
    Sigma: np.ndarray = None
    dist: Normal = None
    history: list = field(default_factory=list)
    gs: list = field(default_factory=list)
    Hs: list = field(default_factory=list)
    metadata: dict = field(default_factory=dict)
    rng: int = None
    time: float = 0.0  # seconds of wall time (the reference keeps Millisecond)

    # the reference's field names
    θ = property(lambda self: self.theta)
    Σ = property(lambda self: self.Sigma)

    def __repr__(self):  # Base.show (src/muse.jl:45-59)
        if self.theta is not None and self.Sigma is not None:
            sig = np.sqrt(np.diag(np.atleast_2d(self.Sigma)))
            body = ", ".join(f"{m:.4g}±{s:.3g}" for m, s in zip(np.atleast_1d(self.theta), sig))
        elif self.theta is not None:
            body = ", ".join(f"{m:.4g}" for m in np.atleast_1d(self.theta))
        else:
            body = ""
        return f"MuseResult({body})"


def _something(*args):
    for a in args:
        if a is not None:
            return a
    raise ValueError("all arguments are None")


def _default_rng():
    return int(np.random.SeedSequence().entropy & ((1 << 63) - 1))


_warned_ignored = set()


def _note_ignored(**kw):
    """`pool` / `progress` exist for signature parity with the reference (src/muse.jl:124-125,305-306,492-493) and do
    nothing here (the map is one GPU launch; there is no progress bar): say so once instead of silently."""
    for name, val in kw.items():
        if val not in (None, False) and name not in _warned_ignored:
            _warned_ignored.add(name)
            warnings.warn(f"`{name}` is accepted for signature parity with MuseInference.jl and ignored: the map over "
                          "simulations is one batched GPU launch", RuntimeWarning, stacklevel=3)


def _has_batch(prob):
    return hasattr(prob, "map_and_score_batch")


# ----------------------------------------------------------------------------------------------
# the map bodies, element by element (LocalWorkerPool path) for problems without batched seams
def _zeros_like(v):
    return v.new_zeros(v.shape) if hasattr(v, "new_zeros") else np.zeros_like(v)   # (a torch tensor stays on its device)


def _as_vector(v):
    return v if hasattr(v, "new_zeros") else np.asarray(v, dtype=np.float64)


def _map_serial(prob, rng, theta, theta_t, sims, include_data, zprev, atol, z0_mode):
    gs, gts, zs, infos = [], [], [], []
    elems = ([None] if include_data else []) + [SimRng(rng, s) for s in sims]
    for e, r in enumerate(elems):
        if r is None:
            x, ztrue = prob.x, None
        else:
            x, ztrue = prob.sample_x_z(r, theta)
        if z0_mode == _capi.Z0_WARM and zprev is not None:
            z0 = zprev[e]
        elif z0_mode == _capi.Z0_TRUE and ztrue is not None:
            z0 = ztrue
        else:
            # zero(sample_x_z(prob, copy(rng), θ).z) (src/muse.jl:146): the latent space need not have the data's shape
            z0 = _zeros_like(ztrue if ztrue is not None else prob.sample_x_z(SimRng(rng, 0), theta)[1])
        zhat, info = prob.zhat_at_theta(x, z0, theta, atol)
        gs.append(np.atleast_1d(prob.grad_theta_logLike(x, zhat, theta, UnTransformedθ)))
        gts.append(np.atleast_1d(prob.grad_theta_logLike(x, zhat, theta_t, Transformedθ)))
        zs.append(zhat)
        infos.append(info)
    return np.array(gs), np.array(gts), zs, infos


# ----------------------------------------------------------------------------------------------
def muse(prob, theta0, **kwargs):
    """muse(prob, θ₀; kwargs...) = muse!(MuseResult(), prob, θ₀; kwargs...)   (src/muse.jl:107)"""
    return muse_(MuseResult(), prob, theta0, **kwargs)


def muse_(result, prob, theta0=None, *, rng=None, z0=None, maxsteps=50, theta_rtol=1e-1,
          grad_z_logLike_atol=1e-2, nsims=100, alpha=0.7, progress=False, pool=None,
          regularize=None, Hinv_like0=None, Hinv_update="sims", broyden_memory=math.inf,
          checkpoint_filename=None, get_covariance=False, save_MAPs=False, native="auto"):
    """muse!(result, prob, θ₀; ...)   (src/muse.jl:112-250).

    Keyword names: θ_rtol -> theta_rtol, ∇z_logLike_atol -> grad_z_logLike_atol, α -> alpha,
    H⁻¹_like′ -> Hinv_like0, H⁻¹_update -> Hinv_update in {"sims","broyden","diagonal_broyden"}.
    `rng` is the master seed; it is stored, never advanced (src/muse.jl:134, src/util.jl:87-92).
    `pool` is accepted for signature parity and ignored: the batch is one GPU launch.
    """
    _note_ignored(pool=pool, progress=progress)
    result.rng = rng = int(_something(rng, result.rng, _default_rng()))
    # `native`: run the whole outer loop in the library's host code (muse_run of the C ABI) when the options are
    # the plain ones it implements -- a fresh run, "sims" Jacobian update, constant alpha, identity regularize,
    # flat/Gaussian prior, untransformed theta, nothing to save per iteration; "auto" = whenever possible.
    plain = (len(result.history) == 0 and regularize is None and Hinv_like0 is None and Hinv_update == "sims"
             and not callable(alpha) and checkpoint_filename is None and save_MAPs is False and nsims >= 2
             and getattr(type(prob), "supports_native_muse", False) and prob.native_prior() is not None)
    if native is True and not plain:
        raise ValueError("native=True needs the plain option set muse_run implements (see muse_hip.h)")
    if native in (True, "auto") and plain:
        return _muse_native(result, prob, _something(result.theta, theta0), rng, z0, maxsteps, theta_rtol,
                            grad_z_logLike_atol, nsims, alpha, get_covariance)
    regularize = (lambda t: t) if regularize is None else regularize
    theta_unreg = theta = prob.standardize_theta(_something(result.theta, theta0))
    theta_unreg_t = theta_t = np.atleast_1d(prob.transform_theta(theta))
    history = result.history
    nth = theta.size
    alpha_fn = alpha if callable(alpha) else (lambda i: alpha)
    save_fn = (lambda z: z) if save_MAPs is True else ((lambda z: None) if save_MAPs is False else save_MAPs)
    batched = _has_batch(prob)
    if z0 is not None and batched:  # starting guess for every element's MAP (src/muse.jl:151)
        prob.set_zhat(0, np.tile(np.asarray(z0, dtype=np.float64), (nsims + 1, 1)))
    zs = None if z0 is None else [_as_vector(z0)] * (nsims + 1)
    Hinv_like = None if Hinv_like0 is None else np.atleast_2d(np.asarray(Hinv_like0, dtype=np.float64))
    first = True  # ẑs = fill(z₀ | zero(z), nsims+1) on every call, resumed or not (src/muse.jl:151)

    for i in range(len(history) + 1, maxsteps + 1):
        t0 = time.perf_counter()
        if i > 2:  # convergence on the last two recorded iterates (src/muse.jl:163-166)
            d = history[-1]["θ′"] - history[-2]["θ′"]
            q = -(d @ history[-1]["H⁻¹_post′"] @ d)
            if q < 0:  # Julia's sqrt of a negative number: DomainError (H⁻¹_post′ not negative definite)
                raise ValueError("DomainError in the convergence test: Δθ′ᵀ H⁻¹_post′ Δθ′ > 0 "
                                 "(H⁻¹_post′ is not negative definite)")
            if math.sqrt(q) < theta_rtol:  # a NaN compares false: the loop goes on, as in the reference
                break
        # MUSE gradient: the (nsims+1)-element map (src/muse.jl:169-181)
        z0_mode = _capi.Z0_WARM if (not first or z0 is not None) else _capi.Z0_ZERO
        first = False
        if batched:
            g, info = prob.map_and_score_batch(rng, 0, nsims, theta, include_data=True,
                                               atol=grad_z_logLike_atol, z0_mode=z0_mode)
            # g′ = ∇θ′ logLike in the transformed space (src/muse.jl:173); identity for untransformed problems
            g, g_t = prob.scores_in_both_spaces(g, theta, theta_t) if hasattr(prob, "scores_in_both_spaces") else (g, g)
            from .problem import check_optim_soln
            check_optim_soln(info, "muse!")
            zhist = info
        else:
            g, g_t, zs, zhist = _map_serial(prob, rng, theta, theta_t, range(nsims), True, zs,
                                            grad_z_logLike_atol, z0_mode)
        g_like_dat, g_like_sims = g[0], g[1:]
        g_like_dat_t, g_like_sims_t = g_t[0], g_t[1:]

        g_like_t = g_like_dat_t - g_like_sims_t.mean(axis=0)                      # :183
        g_prior_t = np.atleast_1d(prob.grad_logPrior_theta(theta_t, Transformedθ))  # :184
        g_post_t = g_like_t + g_prior_t                                              # :185

        # Jacobian (src/muse.jl:188-205)
        Hinv_like_sims = np.diag(-1.0 / np.var(g_like_sims_t, axis=0, ddof=1)) if nsims > 1 else \
            np.diag(np.full(nth, -np.inf))
        if Hinv_like is None or Hinv_update == "sims":
            Hinv_like = Hinv_like_sims
        elif i > 2 and Hinv_update in ("broyden", "diagonal_broyden"):
            j0 = int(max(2, i - broyden_memory))
            Hinv_like = history[j0 - 2]["H⁻¹_like_sims′"]
            for j in range(j0, i):
                dth = history[j - 1]["θ′"] - history[j - 2]["θ′"]
                dg = history[j - 1]["g_like′"] - history[j - 2]["g_like′"]
                Hinv_like = Hinv_like + np.outer((dth - Hinv_like @ dg) / (dth @ Hinv_like @ dg), dth) @ Hinv_like
                if Hinv_update == "diagonal_broyden":
                    Hinv_like = np.diag(np.diag(Hinv_like))
        H_prior_t = np.atleast_2d(prob.hess_logPrior_theta(theta_t, Transformedθ))  # :207
        Hinv_post = np.linalg.inv(np.linalg.inv(Hinv_like) + H_prior_t)                # :208

        t = time.perf_counter() - t0
        history.append({
            "θ": theta, "θunreg": theta_unreg, "θ′": theta_t, "θunreg′": theta_unreg_t,
            "g_like_sims": g_like_sims,
            "g_like_dat′": g_like_dat_t, "g_like_sims′": g_like_sims_t, "g_like′": g_like_t,
            "g_prior′": g_prior_t, "g_post′": g_post_t,
            "H⁻¹_post′": Hinv_post, "H_prior′": H_prior_t, "H⁻¹_like′": Hinv_like,
            "H⁻¹_like_sims′": Hinv_like_sims,
            "ẑ_history_dat": zhist[0], "ẑ_history_sims": zhist[1:], "t": t,
            "ẑ_dat": _saved_map(prob, batched, zs, 0, save_MAPs, save_fn),
            "ẑ_sims": _saved_maps(prob, batched, zs, nsims, save_MAPs, save_fn),
        })

        # Newton-Raphson step (src/muse.jl:224-227)
        theta_unreg_t = theta_t - alpha_fn(i) * (Hinv_post @ g_post_t)
        theta_unreg = np.atleast_1d(prob.inv_transform_theta(theta_unreg_t))
        theta_t = np.atleast_1d(regularize(theta_unreg_t))
        theta = np.atleast_1d(prob.inv_transform_theta(theta_t))

        result.theta = theta_unreg      # :230
        result.gs = list(g_like_sims)   # :231
        result.time += t                # :232
        if checkpoint_filename is not None:
            save_result(checkpoint_filename, result)

    if get_covariance:
        get_J_(result, prob, rng=rng, nsims=nsims, grad_z_logLike_atol=grad_z_logLike_atol)
        get_H_(result, prob, rng=rng, nsims=max(1, nsims // 10), grad_z_logLike_atol=grad_z_logLike_atol)
    return result


def _muse_native(result, prob, theta0, rng, z0, maxsteps, theta_rtol, atol, nsims, alpha, get_covariance):
    """muse! through muse_run: the same history records as muse_ builds, from the arrays the library returns."""
    from .problem import check_optim_soln
    theta0 = prob.standardize_theta(theta0)
    nth = theta0.size
    if z0 is not None:  # starting guess for every element's MAP (src/muse.jl:151)
        prob.set_zhat(0, np.tile(np.asarray(z0, dtype=np.float64), (nsims + 1, 1)))
    n, theta, hist, gs, info = prob.run_muse(rng, theta0, nsims=nsims, maxsteps=maxsteps, theta_rtol=theta_rtol,
                                             atol=atol, alpha=float(alpha), z0_warm=z0 is not None)
    check_optim_soln(info[:n].reshape(-1), "muse!")   # (once for the run: 30 iterations x 513 records are one vector op)
    for i in range(n):
        h = hist[i]                                      # (records are views of the arrays the library filled: no copies)
        th = h[0:nth]
        seg = lambda k: h[k * nth:(k + 1) * nth]
        Hinv_like = np.diag(seg(5))
        result.history.append({
            "θ": th, "θunreg": th, "θ′": th, "θunreg′": th,
            "g_like_sims": gs[i], "g_like_dat′": seg(1), "g_like_sims′": gs[i], "g_like′": seg(2),
            "g_prior′": seg(3), "g_post′": seg(4),
            "H⁻¹_post′": h[7 * nth:7 * nth + nth * nth].reshape(nth, nth), "H_prior′": np.diag(seg(6)),
            "H⁻¹_like′": Hinv_like, "H⁻¹_like_sims′": Hinv_like,
            "ẑ_history_dat": info[i][0], "ẑ_history_sims": info[i][1:], "t": float(h[-1]),
            "ẑ_dat": None, "ẑ_sims": [None] * nsims,
        })
        result.time += float(h[-1])
    result.theta = theta
    result.gs = list(gs[n - 1])
    if get_covariance:
        get_J_(result, prob, rng=rng, nsims=nsims, grad_z_logLike_atol=atol)
        get_H_(result, prob, rng=rng, nsims=max(1, nsims // 10), grad_z_logLike_atol=atol)
    return result


def _saved_map(prob, batched, zs, slot, save_MAPs, save_fn):
    if save_MAPs is False:
        return None
    return save_fn(prob.get_zhat(slot, slot + 1)[0] if batched else zs[slot])


def _saved_maps(prob, batched, zs, nsims, save_MAPs, save_fn):
    if save_MAPs is False:
        return [None] * nsims
    if batched:
        Z = prob.get_zhat(1, nsims + 1)
        return [save_fn(Z[k]) for k in range(nsims)]
    return [save_fn(z) for z in zs[1:]]


# ----------------------------------------------------------------------------------------------
def get_J_(result, prob, theta0=None, *, z0=None, grad_z_logLike_atol=1e-2, rng=None, nsims=100, pool=None,
           progress=False, skip_errors=False, covariance_method="simple_corrected"):
    """get_J!(result, prob, θ₀; ...)   (src/muse.jl:484-532).  J = var(gs) / corrected sample covariance
    (SimpleCovariance(corrected=true), src/muse.jl:495,529); only nsims - length(result.gs) new sims are
    run, continuing the same streams (src/muse.jl:499-506).  covariance_method: the estimator applied to the scores
    (src/muse.jl:480: any CovarianceEstimator) -- covariance.SimpleCovariance / LinearShrinkage, a callable scores -> J, or a name."""
    from .covariance import as_covariance_method
    estimator = as_covariance_method(covariance_method)       # (refused before any simulation runs)
    _note_ignored(pool=pool, progress=progress)
    rng = int(_something(rng, result.rng, _default_rng()))
    theta0 = prob.standardize_theta(_something(theta0, result.theta))
    existing = len(result.gs)
    if nsims - existing > 0:
        if _has_batch(prob) and z0 is None:
            g, info = prob.map_and_score_batch(rng, existing, nsims, theta0, include_data=False,
                                               atol=grad_z_logLike_atol, z0_mode=_capi.Z0_TRUE)
            if hasattr(prob, "scores_in_both_spaces"):  # J is built from untransformed-space scores (src/muse.jl:513)
                g, _ = prob.scores_in_both_spaces(g, theta0, prob.transform_theta(theta0))
            g, _ = _apply_skip_errors(g, info, skip_errors, "get_J!")
        else:
            zst = None if z0 is None else [_as_vector(z0)] * (nsims - existing)
            g, _, _, _ = _map_serial(prob, rng, theta0, theta0, range(existing, nsims), False, zst,
                                     grad_z_logLike_atol, _capi.Z0_WARM if z0 is not None else _capi.Z0_TRUE)
        result.gs = list(result.gs) + list(g)
    G = np.array(result.gs)
    result.J = np.atleast_2d(np.asarray(estimator(G), dtype=np.float64))
    return finalize_result_(result, prob)


def _apply_skip_errors(g, info, skip_errors, where):
    """skip_errors semantics (src/muse.jl:515-521): a failed element is dropped (`missing`) when
    skip_errors, otherwise the failure propagates.  Non-convergence alone is a warning."""
    from .problem import check_optim_soln
    flat = np.asarray(info).reshape(-1)
    bad = flat["status"] == _capi.STATUS_NONFINITE
    check_optim_soln(flat, where)
    if np.any(bad):
        if not skip_errors:
            raise FloatingPointError(f"{where}: {int(bad.sum())} MAP solve(s) ended non-finite")
        keep = ~bad.reshape(np.asarray(info).shape).reshape(len(g), -1).any(axis=1)
        return g[keep], keep
    return g, np.ones(len(g), dtype=bool)


def get_H_(result, prob, theta0=None, *, fdm="central_fdm(3,1)", grad_z_logLike_atol=1e-2, rng=None, nsims=10,
           step=None, pool=None, pmap_over="auto", progress=False, skip_errors=False, z0=None,
           implicit_diff=False, implicit_diff_H1_is_zero=False, implicit_diff_cg_kwargs=None, fid_mode=0):
    """get_H!(result, prob, θ₀; ...) finite-difference branch   (src/muse.jl:296-333, 407-450).

    fdm: "central_fdm(p,1)" or a fdm.FiniteDifferenceMethod (reference default central_fdm(3,1), src/muse.jl:300).
    step defaults to 0.1 ./ std(result.gs) (src/muse.jl:411-413); with neither a step nor result.gs the method estimates
    its own step per simulation and column, as FiniteDifferences does (fdm.py).  fid_mode 0 reproduces the reference's
    fiducial warm start (every FD MAP starts from the MAP of the one simulation drawn from the un-split
    master stream, src/muse.jl:417-423); fid_mode 1 uses each sim's own fiducial MAP.
    """
    _note_ignored(pool=pool, progress=progress)
    if implicit_diff:
        return _get_H_implicit(result, prob, theta0, rng, nsims, implicit_diff_cg_kwargs, skip_errors)
    method = as_fdm(fdm)
    if method.q != 1:
        raise ValueError("get_H! differentiates once: fdm must be a first-derivative method (central_fdm(p, 1))")
    rng = int(_something(rng, result.rng, _default_rng()))
    theta0 = prob.standardize_theta(_something(theta0, result.theta))
    existing = len(result.Hs)
    remaining = nsims - existing
    if remaining <= 0:
        return result
    t0 = time.perf_counter()
    if step is None and len(result.gs) > 0:
        step = 0.1 / np.std(np.array(result.gs), axis=0, ddof=1)
    if step is not None:
        step = np.broadcast_to(np.atleast_1d(np.asarray(step, dtype=np.float64)), theta0.shape).copy()
    # split_rng(rng, nsims_remaining): streams 0 .. remaining-1 (src/muse.jl:323)
    Hs = None
    if _has_batch(prob) and z0 is None:
        try:
            if step is not None and method.grid == [-1, 0, 1]:   # the reference default with an explicit step: one entry
                Hs, info = prob.fd_jacobian_batch(rng, 0, remaining, theta0, step, atol=grad_z_logLike_atol,
                                                  fid_mode=fid_mode, fid_sim=MASTER_SIM)
            elif hasattr(prob, "fd_values_columns"):
                Hs, info = _fd_batched(prob, rng, remaining, theta0, method, step, grad_z_logLike_atol, fid_mode)
            else:
                raise NotImplementedError
            Hs, _ = _apply_skip_errors(Hs, info, skip_errors, "get_H!")
        except NotImplementedError:
            Hs = None
    if Hs is None:
        Hs = _fd_serial(prob, rng, remaining, theta0, method, step, grad_z_logLike_atol, z0, fid_mode)
    result.Hs = list(result.Hs) + list(Hs)
    result.H = np.mean(np.array(result.Hs), axis=0)
    result.time += time.perf_counter() - t0
    return finalize_result_(result, prob)


def _fd_batched(prob, rng, nsims, theta0, m, step, atol, fid_mode):
    """get_H!'s finite-difference map for any central_fdm(p, 1), with an explicit step or -- neither `step` nor result.gs --
    FiniteDifferences' own step estimation (src/muse.jl:300,411-413; src/util.jl:13: fdm(f, 0) without a step), through the
    engine's raw-value seam: every grid point of every (simulation, column) unit is one MAP + score problem of ONE launch.
    The step is estimated per fdm call, i.e. per simulation and column, from the method's bound estimator (order p + 2)
    evaluated at ITS default step."""
    nth = theta0.size
    n = nsims * nth
    kw = dict(atol=atol, fid_mode=fid_mode, fid_sim=MASTER_SIM)
    grid = np.array(m.grid, dtype=np.float64)
    infos = []
    if step is not None:
        nz = np.flatnonzero(m.coefs != 0.0)                     # a point with coefficient 0 is not evaluated
        F, info = prob.fd_values_columns(rng, 0, 0, n, theta0, step[:, None] * grid[None, nz], **kw)
        cols = np.tensordot(F, m.coefs[nz], axes=(1, 0)) / np.tile(step, nsims)[:, None] ** m.q
        infos.append(info)
    else:
        est = m.bound_estimator
        if est is None or est.bound_estimator is not None:
            raise NotImplementedError("batched step estimation is built for adapt = 1")
        step_e = est.estimate_step(None)
        Fe, info_e = prob.fd_values_columns(rng, 0, 0, n, theta0, np.tile(step_e * np.array(est.grid, dtype=np.float64), (nth, 1)), **kw)
        steps = np.array([m.step_from_magnitudes(*est.magnitudes_from_values(Fe[u], step_e)) for u in range(n)])
        F, info = prob.fd_values_columns(rng, 0, 0, n, theta0, steps[:, None] * grid[None, :], per_unit=True, **kw)
        cols = np.tensordot(F, m.coefs, axes=(1, 0)) / steps[:, None] ** m.q
        infos += [info_e, info]
    # the per-sim Jacobian is the hcat of its columns (src/util.jl:25): Hs[s][i][j] = cols[s*nθ + j][i]
    Hs = np.ascontiguousarray(cols.reshape(nsims, nth, nth).transpose(0, 2, 1))
    info = np.concatenate([i.reshape(nsims, -1) for i in infos], axis=1)
    return Hs, info


def _get_H_implicit(result, prob, theta0, rng, nsims, cg_kwargs, skip_errors):
    """get_H! with implicit_diff=true (src/muse.jl:335-405): H = H1 - dFdθᵀ A⁻¹ dFdθ1 per sim, A⁻¹ by CG
    (implicit_diff_cg_kwargs default (maxiter=100, Pl=I)); the fiducial MAP is solved to 1e-1 as the
    reference hard-codes (src/muse.jl:344).  CG iteration counts go to metadata["implicit_diff_cg_hists"]."""
    rng = int(_something(rng, result.rng, _default_rng()))
    theta0 = prob.standardize_theta(_something(theta0, result.theta))
    remaining = nsims - len(result.Hs)
    if remaining <= 0:
        return result
    if not hasattr(prob, "implicit_H_batch"):
        raise NotImplementedError("implicit_diff needs a problem with the implicit_H_batch seam")
    t0 = time.perf_counter()
    maxiter = int((cg_kwargs or {}).get("maxiter", 100))
    Hs, its = prob.implicit_H_batch(rng, 0, remaining, theta0, atol=1e-1, cg_maxiter=maxiter)
    result.Hs = list(result.Hs) + list(Hs)
    result.metadata.setdefault("implicit_diff_cg_hists", []).extend(list(its))
    result.H = np.mean(np.array(result.Hs), axis=0)
    result.time += time.perf_counter() - t0
    return finalize_result_(result, prob)


def _fd_serial(prob, rng, nsims, theta0, m, step, atol, z0, fid_mode):
    """The same map element by element through the per-simulation operators (pjacobian, src/util.jl:9-27)."""
    nth = theta0.size
    Hs = []
    zfid_master = None
    for s in range(nsims):
        if fid_mode == 0 and zfid_master is None or fid_mode == 1:
            x, z = prob.sample_x_z(SimRng(rng, MASTER_SIM if fid_mode == 0 else s), theta0)
            zs = prob.zhat_guess_from_truth(x, z, theta0) if z0 is None else z0
            zfid, _ = prob.zhat_at_theta(x, zs, theta0, atol)
            zfid_master = zfid
        zfid = zfid_master
        H = np.empty((nth, nth))
        for j in range(nth):
            def f(eps, j=j, s=s, zfid=zfid):
                th = theta0.copy()
                th[j] = theta0[j] + eps
                x, _ = prob.sample_x_z(SimRng(rng, s), th)
                zh, _ = prob.zhat_at_theta(x, zfid, theta0, atol)
                return np.atleast_1d(prob.grad_theta_logLike(x, zh, theta0, UnTransformedθ))
            if step is not None and m.grid == [-1, 0, 1]:
                H[:, j] = (-0.5 * f(-step[j]) + 0.5 * f(step[j])) / step[j]
            else:
                H[:, j] = m(f, 0.0, None if step is None else step[j])
        Hs.append(H)
    return Hs


def finalize_result_(result, prob):
    """finalize_result!   (src/muse.jl:535-549):  Σ⁻¹ = Hᵀ J⁻¹ H + H_prior,  H_prior = -∇²θ logPrior(θ̂)."""
    if result.H is not None and result.J is not None and result.theta is not None:
        H = np.atleast_2d(result.H)
        J = np.atleast_2d(result.J)
        H_prior = -np.atleast_2d(prob.hess_logPrior_theta(result.theta, UnTransformedθ))
        result.Sigma_inv = H.T @ np.linalg.inv(J) @ H + H_prior
        result.Sigma = np.linalg.inv(result.Sigma_inv)
        result.dist = Normal(np.atleast_1d(result.theta), 0.5 * (result.Sigma + result.Sigma.T))
    return result


# ----------------------------------------------------------------------------------------------
def save_result(filename, result):
    """save(checkpoint_filename, "result", result)   (src/muse.jl:234)"""
    with open(filename, "wb") as f:
        pickle.dump(result, f)


def load_result(filename):
    with open(filename, "rb") as f:
        return pickle.load(f)
