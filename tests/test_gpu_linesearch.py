"""The kernel's HagerZhang line search on a NON-quadratic objective.

Every shipped model is Gaussian, i.e. quadratic in z, and on a quadratic the line search ends with its first secant step:
bracket expansion (B1-B3), bisection (U3), the secant^2 updates (S1-S4) and the interval-shrink test of the main loop are
code the product's own maps never reach.  A diagnostic build (-DMUSE_HZTEST, museinference.jl_amd/csrc/models.hpp: the noise
model's objective becomes 1/2 z^2 + 1/2 e^-theta (x - z)^2 + 1/4 z^4; one solver instantiation per library, built in
seconds next to the product library, never loaded by the product path) runs the kernel's solver on such an objective in
all three storage policies, and the oracle's solver -- the restatement that reproduces the Optim.jl documentation's
Rosenbrock counters (tests/test_oracle.py) -- solves the same problems through its test objective 101: 8-30 L-BFGS
iterations with 2.7-2.9 evaluations each.  Equal iteration and evaluation counts, MAPs to 1e-9, minima to 1e-12 where the
paths agree; where a count differs (tree-ordered against sequential sums over 10^4-10^5 terms) both must
have converged to the same MAP within the solve's tolerance (the objective is strictly convex: lambda_min >= 1)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("kind", ["resident", "streaming", "cluster"])
def test_kernel_line_search_on_a_non_quadratic_objective(gpu, M, O, kind, tmp_path):
    from museinference_jl_amd import build
    lib = build.build_linesearch_test_variants()[kind]          # (seconds; normally built already by __graft_entry__.build())
    out = str(tmp_path / "res.npz")
    p = subprocess.run([sys.executable, os.path.join(HERE, "linesearch_worker.py"), kind, out], env=dict(os.environ, MUSE_HIP_LIB=lib),
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = np.load(out)
    n = int(d["ncases"])
    assert n == 12
    same_path = 0
    iters, report = [], []
    for c in range(n):
        theta, atol, f_min, gnorm = d[f"par{c}"]
        it, fc, status = (int(v) for v in d[f"info{c}"])
        zo, io = O.zhat_at_theta("quartic_test", d[f"x{c}"], d[f"z0{c}"], [theta], atol)
        report.append((c, theta, atol, (it, fc, status), (io["iterations"], io["f_calls"], io["status"])))
        # converged one way or another on both sides (0: gradient, 1: zero step, 2: objective unchanged twice -- what a 1e-7
        # tolerance on 10^4-10^5 elements often ends with), and the same way
        assert status in (0, 1, 2) and io["status"] in (0, 1, 2), report[-1]
        iters.append(io["iterations"])
        if (it, fc, status) == (io["iterations"], io["f_calls"], io["status"]):
            same_path += 1
            np.testing.assert_allclose(d[f"z{c}"], zo, rtol=0, atol=1e-9, err_msg=f"{kind} case {c}")
            np.testing.assert_allclose(f_min, io["f_min"], rtol=1e-12)
        else:   # the strictly convex objective has one MAP: both ended within the solve's accuracy of it
            # (an early stop by "objective unchanged twice" -- sums of 10^4-10^5 terms in another order -- saves a few iterations)
            assert abs(it - io["iterations"]) <= (3 if status == io["status"] else 8), report[-1]
            assert np.abs(d[f"z{c}"] - zo).max() <= max(2 * atol, 1e-6), report[-1]
        if status == 0:
            assert gnorm <= atol
    print(kind, report)
    assert max(iters) >= 15 and min(iters) >= 3          # real line-search work, not one secant step
    assert same_path >= n - 3, (kind, same_path, report)
