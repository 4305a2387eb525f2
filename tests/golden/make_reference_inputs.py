#!/usr/bin/env python3
"""Writes tests/golden/reference_inputs.npz: the inputs julia/make_reference_fixtures.jl feeds to the REAL
MuseInference.jl (MAP cases: x, z0, theta, atol for the three models; one muse! run: data, the standard normals of the
master stream and of every simulation, options).  The Julia script's output, tests/golden/reference_outputs.npz, is what
tests/test_reference_fixtures.py checks the oracle against -- the pin that does not come from this repository's own
arithmetic (SURVEY.md §8 c).  Run from the repository root:  python tests/golden/make_reference_inputs.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
MODEL_ID = {"funnel": 0, "noise": 1, "smooth": 2}
DATA_SIM, MASTER_SIM = (1 << 32) - 1, (1 << 62) - 1

CASES = [  # (model, N, ntheta, theta, atol, start)
    ("funnel", 512, 1, [1.0], 1e-2, "zero"), ("funnel", 512, 1, [-0.5], 1e-6, "zero"), ("funnel", 300, 3, [0.5, -0.5, 1.5], 1e-2, "zero"),
    ("noise", 512, 1, [0.3], 1e-2, "zero"), ("noise", 257, 1, [-0.4], 1e-6, "true"),
    ("smooth", 600, 4, [1.0, 2.0, 3.0, 0.5], 1e-2, "zero"), ("smooth", 600, 4, [1.0, 2.0, 3.0, 0.5], 1e-6, "zero"),
    ("smooth", 128, 2, [2.0, 3.0], 1e-5, "true"),
]


def main():
    O.build()
    out = {"ncases": np.array(len(CASES))}
    for c, (model, N, nth, theta, atol, start) in enumerate(CASES):
        x, z = O.sample_x_z(model, N, 7, c, theta)
        out.update({f"case{c}_model": np.array(MODEL_ID[model]), f"case{c}_N": np.array(N), f"case{c}_ntheta": np.array(nth),
                    f"case{c}_theta": np.array(theta), f"case{c}_atol": np.array(atol), f"case{c}_x": x,
                    f"case{c}_z0": np.zeros(N) if start == "zero" else z})
    # one muse! run: the 512-dim funnel of the reference's own tests (test/runtests.jl:12-37), nsims = 16
    N, nsims, seed = 512, 16, 42
    x, _ = O.sample_x_z("funnel", N, 123, DATA_SIM, [0.0])
    n1 = np.empty((nsims + 1, N))
    n2 = np.empty((nsims + 1, N))
    n1[0], n2[0] = O.normals(seed, MASTER_SIM, N)            # row 0: the un-split master stream (get_H!'s fiducial, src/muse.jl:418)
    for k in range(nsims):
        n1[k + 1], n2[k + 1] = O.normals(seed, k, N)
    out.update({"run_model": np.array(0), "run_N": np.array(N), "run_ntheta": np.array(1), "run_nsims": np.array(nsims),
                "run_seed": np.array(seed), "run_x": x, "run_n1": n1, "run_n2": n2, "run_theta0": np.array([1.0]),
                "run_prior_sigma": np.array(3.0), "run_maxsteps": np.array(50), "run_theta_rtol": np.array(1e-1),
                "run_atol": np.array(1e-2), "run_alpha": np.array(0.7)})
    np.savez_compressed(os.path.join(HERE, "reference_inputs.npz"), **out)
    print("wrote reference_inputs.npz")


if __name__ == "__main__":
    main()
