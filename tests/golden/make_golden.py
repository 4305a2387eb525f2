#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (oracle/muse_oracle.c).

The reference (Julia) cannot run in this image and ships no golden vectors for this path
(SURVEY.md §8c), so these fixtures are outputs of the build's own oracle for fixed Philox seeds; the
oracle itself is pinned against closed forms, scipy and the Philox known-answer vectors in
tests/test_oracle.py.  Fixtures: for seeds {0,1,2} and N in {8,512}: x, z, zhat, score, per-sim
finite-difference H; an L-BFGS trace on the non-isotropic (smooth) model; a full muse_ trajectory.
Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402

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


def trajectory():
    import museinference_jl_amd as M
    from oracle_problem import OracleBatchedProblem
    N = 512
    x, _ = O.sample_x_z("funnel", N, 123, M.DATA_SIM, [0.0])
    prob = OracleBatchedProblem(x, "funnel", 1, prior=M.GaussianPrior(0.0, 3.0))
    res = M.muse(prob, [1.0], rng=42, nsims=32, get_covariance=True)
    np.savez_compressed(os.path.join(HERE, "muse_trajectory.npz"), x=x,
                        thetas=np.array([h["θ"] for h in res.history]),
                        g_like=np.array([h["g_like′"] for h in res.history]),
                        Hinv_post=np.array([h["H⁻¹_post′"] for h in res.history]),
                        theta=res.theta, J=res.J, H=res.H, Sigma=res.Sigma, gs=np.array(res.gs))


if __name__ == "__main__":
    O.build()
    per_sim()
    trajectory()
    print("wrote", sorted(os.listdir(HERE)))
