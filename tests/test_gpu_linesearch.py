"""The kernel's HagerZhang line search on a NON-quadratic objective.

Every built-in model is Gaussian, i.e. quadratic in z, and on a quadratic the line search ends with its first secant step:
bracket expansion (B1-B3), bisection (U3), the secant^2 updates (S1-S4) and the interval-shrink test of the main loop are
code the built-in models' maps never reach.  tests/models/quartic.h -- a user-supplied model (include/muse_model.h) with the
objective 1/2 z^2 + 1/4 z^4 + 1/2 e^-theta (x - z)^2, compiled into an engine library of its own and into the oracle's
checker build -- runs the kernel's solver on such an objective in all three storage policies, and the oracle's solver -- the
restatement that reproduces the Optim.jl documentation's Rosenbrock counters (tests/test_oracle.py) -- solves the same
problems: 8-30 L-BFGS iterations with 2.7-2.9 evaluations each.  Equal iteration and evaluation counts, MAPs to 1e-9,
minima to 1e-12 where the paths agree; where a count differs (tree-ordered against sequential sums over 10^4-10^5 terms,
DESIGN.md §5) both must have converged to the same MAP within the solve's tolerance (the objective is strictly convex:
lambda_min >= 1)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
QUARTIC = os.path.join(HERE, "models", "quartic.h")


@pytest.mark.parametrize("kind", ["resident", "streaming", "cluster"])
def test_kernel_line_search_on_a_non_quadratic_objective(gpu, M, O, kind):
    N = {"resident": 10000, "streaming": 7001, "cluster": 70000}[kind]
    prob = M.HipMuseProblem(None, model=M.ElementwiseModel("quartic", QUARTIC), ntheta=1, N=N)
    if kind == "streaming":
        prob.set_placement(0)
    pi = prob.placement_info()
    assert pi["resident"] == (kind == "resident") and (pi["workgroups_per_element"] > 1) == (kind == "cluster")
    rng = np.random.default_rng(11)
    same_path, n = 0, 0
    iters, report = [], []
    with O.user_model(QUARTIC, "quartic"):
        for theta in (-2.0, 0.0, 1.5):
            for scale, start in ((3.0, "zero"), (1.0, "far")):
                for atol in (1e-2, 1e-7):
                    x = rng.standard_normal(N) * scale
                    z0 = np.zeros(N) if start == "zero" else 4.0 * rng.standard_normal(N)
                    z, info = prob.zhat_at_theta(x, z0, [theta], atol)
                    it, fc, status = info["iterations"], info["f_calls"], info["status"]
                    zo, io = O.zhat_at_theta("user", x, z0, [theta], atol)
                    report.append((n, theta, atol, (it, fc, status), (io["iterations"], io["f_calls"], io["status"])))
                    n += 1
                    # converged one way or another on both sides (0: gradient, 1: zero step, 2: objective unchanged twice --
                    # what a 1e-7 tolerance on 10^4-10^5 elements often ends with), and the same way
                    assert status in (0, 1, 2) and io["status"] in (0, 1, 2), report[-1]
                    iters.append(io["iterations"])
                    if (it, fc, status) == (io["iterations"], io["f_calls"], io["status"]):
                        same_path += 1
                        np.testing.assert_allclose(z, zo, rtol=0, atol=1e-9, err_msg=f"{kind} case {n}")
                        np.testing.assert_allclose(info["f_min"], io["f_min"], rtol=1e-12)
                    else:   # the strictly convex objective has one MAP: both ended within the solve's accuracy of it (an early
                        # stop by "objective unchanged twice" -- sums of 10^4-10^5 terms in another order -- saves a few iterations)
                        assert abs(it - io["iterations"]) <= (3 if status == io["status"] else 8), report[-1]
                        assert np.abs(z - zo).max() <= max(2 * atol, 1e-6), report[-1]
                    if status == 0:
                        assert info["gnorm"] <= atol
    prob.close()
    print(kind, report)
    assert n == 12 and max(iters) >= 15 and min(iters) >= 3          # real line-search work, not one secant step
    assert same_path >= n - 3, (kind, same_path, report)
