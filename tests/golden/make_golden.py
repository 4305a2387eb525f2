#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (oracle/muse_oracle.c).

The reference (Julia) cannot run in this image and ships no golden vectors for this path
(SURVEY.md §8c), so these fixtures are outputs of the build's own oracle for fixed Philox seeds; the
oracle itself is pinned against closed forms, scipy and the Philox known-answer vectors in
tests/test_oracle.py.  Fixtures: for seeds {0,1,2} and N in {8,512}: x, z, zhat, score, per-sim
finite-difference H; an L-BFGS trace on the non-isotropic (smooth) model; full muse!/get_J!/get_H!
trajectories whose host algebra comes from tests/muse_reference.py -- a restatement of src/muse.jl that
shares nothing with the product package (this script does not import museinference.jl_amd).
Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
import muse_reference as R  # noqa: E402  (tests/muse_reference.py: independent of the product package)

HERE = os.path.dirname(os.path.abspath(__file__))


def per_sim():
    out = {}
    for model, nth, theta in [("funnel", 1, [1.0]), ("funnel", 2, [0.5, -0.5]), ("noise", 1, [0.3]),
                              ("smooth", 2, [1.0, 2.0])]:
        for N in (8, 512):
            for seed in (0, 1, 2):
                key = f"{model}{nth}_N{N}_s{seed}"
                x, z = O.sample_x_z(model, N, seed, 5, theta)
                zh, info = O.zhat_at_theta(model, x, np.zeros(N), theta, 1e-2)
                zh6, info6 = O.zhat_at_theta(model, x, np.zeros(N), theta, 1e-6)
                s = O.grad_theta(model, x, zh, theta)
                f, g = O.logLike_and_grad_z(model, x, 0.5 * z, theta)
                step = np.full(nth, 0.05)
                _, zfid, _ = O.map_and_score_batch(model, N, seed, 7, 8, theta, atol=1e-2, z0_mode=0)
                H = O.fd_jacobian(model, N, seed, 5, theta, step, zfid[0], atol=1e-2)
                out.update({key + "_x": x, key + "_z": z, key + "_zhat": zh, key + "_zhat6": zh6, key + "_score": s,
                            key + "_iters": np.array([info["iterations"], info["f_calls"], info6["iterations"],
                                                      info6["f_calls"]]),
                            key + "_logLike": np.array([f]), key + "_gradz": g, key + "_H": H,
                            key + "_theta": np.array(theta)})
    np.savez_compressed(os.path.join(HERE, "per_sim.npz"), **out)


DATA_SIM = (1 << 32) - 1    # stream of the synthetic "observed" data
MASTER_SIM = (1 << 62) - 1  # stream standing for the un-split master rng (src/muse.jl:418)


class OracleMap:
    """The pmap body of muse! (src/muse.jl:169-176) on the CPU oracle: element 0 = the data, then sims 0..nsims-1;
    MAPs start from zero on the first call and from the previous call's MAPs afterwards (src/muse.jl:151,181).
    `to_engine` maps the user's theta to the engine's (log-variance) theta; `scores` turns the engine's score
    d logLike / d theta_engine into (g untransformed, g' transformed)."""

    def __init__(self, model, x, nth, seed, nsims, atol=1e-2, to_engine=None, scores=None):
        self.model, self.x, self.nth, self.seed, self.nsims, self.atol = model, x, nth, seed, nsims, atol
        self.to_engine = to_engine or (lambda t: t)
        self.scores = scores or (lambda g, theta, theta_t: (g, g))
        self.zhat = None

    def __call__(self, i, theta, theta_t):
        mode = 0 if self.zhat is None else 2
        g, zh, info = O.map_and_score_batch(self.model, self.x.size, self.seed, 0, self.nsims,
                                            self.to_engine(np.asarray(theta)), atol=self.atol, x_data=self.x,
                                            z0_mode=mode, zhat=self.zhat, nthreads=4)
        self.zhat = zh
        gu, gt = self.scores(g, np.asarray(theta), np.asarray(theta_t))
        return [list(r) for r in gu], [list(r) for r in gt]


def covariance(model, x, nth, seed, theta_hat, gs, nsims_H, atol=1e-2):
    """get_J! on the scores muse! left behind (src/muse.jl:499-502,529) and get_H!'s finite-difference branch
    (src/muse.jl:407-446) from the oracle's per-simulation operators: fiducial MAP of the master stream from
    zero(z) (:417-423), then per sim and column the central difference of the score at theta0 of the MAP (from
    the fiducial MAP) of the simulation re-drawn at theta0 +- step e_j with the same randoms (:428-433)."""
    N = x.size
    J = R.J_from_scores(gs)
    step = R.fd_step_from_scores(gs)
    th0 = np.asarray(theta_hat, dtype=np.float64)
    xm, _ = O.sample_x_z(model, N, seed, MASTER_SIM, th0)
    zfid, _ = O.zhat_at_theta(model, xm, np.zeros(N), th0, atol)
    cols = []
    for s in range(nsims_H):
        cj = []
        for j in range(nth):
            f = []
            for sgn in (+1.0, -1.0):
                th = th0.copy()
                th[j] += sgn * step[j]
                xs, _ = O.sample_x_z(model, N, seed, s, th)
                zh, _ = O.zhat_at_theta(model, xs, zfid, th0, atol)
                f.append(list(O.grad_theta(model, xs, zh, th0)))
            cj.append(R.central_fdm_3_1(f[0], f[1], step[j]))
        cols.append(cj)
    Hs, H = R.H_from_columns(cols)
    return J, Hs, H, step


def gaussian_prior(sigma):
    return (lambda t: [-v / sigma**2 for v in t]), (lambda t: [[-1.0 / sigma**2 if a == b else 0.0 for b in range(len(t))]
                                                               for a in range(len(t))])


def trajectory():
    """muse! + get_J! + get_H! + finalize_result! by tests/muse_reference.py (an independent restatement of the
    reference's host algebra) on the oracle's map: the 512-dim funnel of the reference's tests."""
    N = 512
    x, _ = O.sample_x_z("funnel", N, 123, DATA_SIM, [0.0])
    pg, ph = gaussian_prior(3.0)
    hist, theta, gs = R.muse_loop(OracleMap("funnel", x, 1, 42, 32), [1.0], nsims=32, prior_grad_t=pg, prior_hess_t=ph)
    J, Hs, H, step = covariance("funnel", x, 1, 42, theta, gs, 3)
    _, Sigma = R.finalize(H, J, ph(theta))
    np.savez_compressed(os.path.join(HERE, "muse_trajectory.npz"), x=x,
                        thetas=np.array([h["θ"] for h in hist]),
                        g_like=np.array([h["g_like′"] for h in hist]),
                        Hinv_post=np.array([h["H⁻¹_post′"] for h in hist]),
                        theta=np.array(theta), J=np.array(J), H=np.array(H), Sigma=np.array(Sigma), gs=np.array(gs),
                        Hs=np.array(Hs), step=np.array(step))


def outer_loop_variants():
    """Rows f2/f4: the keyword variants of muse! -- Broyden / diagonal-Broyden Jacobian updates with and without a
    memory limit and a user H^-1_like' (src/muse.jl:192-205), callable alpha and `regularize` (:145-149,224-227),
    and a transformed theta space (variances v = e^theta: theta' = log v) -- on a 4-block funnel, N = 400."""
    N, nth, nsims, seed = 400, 4, 24, 1
    x, _ = O.sample_x_z("funnel", N, 9, DATA_SIM, [0.0] * nth)
    pg, ph = gaussian_prior(3.0)
    out = {"x": x}
    variants = {
        "sims": dict(),
        "broyden": dict(Hinv_update="broyden"),
        "diagonal_broyden": dict(Hinv_update="diagonal_broyden"),
        "broyden_mem2": dict(Hinv_update="broyden", broyden_memory=2),
        "broyden_H0": dict(Hinv_update="broyden", Hinv_like0=[[-0.02 if a == b else 0.0 for b in range(nth)] for a in range(nth)]),
        "alpha_regularize": dict(alpha=lambda i: 1.0 / (1 + i), regularize=lambda t: [min(max(v, -0.5), 0.8) for v in t]),
    }
    for name, kw in variants.items():
        hist, theta, gs = R.muse_loop(OracleMap("funnel", x, nth, seed, nsims), [1.0] * nth, nsims=nsims, prior_grad_t=pg,
                                      prior_hess_t=ph, maxsteps=7, theta_rtol=0.0, **kw)
        out[name + "_thetas"] = np.array([h["θ"] for h in hist] + [theta])
        out[name + "_Hinv_like"] = np.array([h["H⁻¹_like′"] for h in hist])
        out[name + "_Hinv_post"] = np.array([h["H⁻¹_post′"] for h in hist])
        out[name + "_g_post"] = np.array([h["g_post′"] for h in hist])
    # transformed space: the user's theta are variances v; the engine's score is d logLike / d log v
    #   g = score / v (untransformed), g' = score + 1 (transformed, with the log-volume term);
    #   prior on v: log v ~ N(0, 3^2) with density in v  =>  logPrior'(t) = -t^2/18 - t + t = -t^2/18
    scores = lambda g, v, t: (g / v[None, :], g + 1.0)
    hist, v_hat, gs = R.muse_loop(OracleMap("funnel", x, nth, seed, nsims, to_engine=np.log, scores=scores),
                                  [np.e] * nth, nsims=nsims, prior_grad_t=pg, prior_hess_t=ph, maxsteps=5, theta_rtol=0.0,
                                  transform=lambda v: [np.log(a) for a in v], inv_transform=lambda t: [np.exp(a) for a in t])
    out["positive_thetas"] = np.array([h["θ"] for h in hist] + [v_hat])
    out["positive_gs"] = np.array(gs)
    out["positive_g_like_t"] = np.array([h["g_like′"] for h in hist])
    np.savez_compressed(os.path.join(HERE, "muse_outer_variants.npz"), **out)


if __name__ == "__main__":
    O.build()
    per_sim()
    trajectory()
    outer_loop_variants()
    print("wrote", sorted(os.listdir(HERE)))
