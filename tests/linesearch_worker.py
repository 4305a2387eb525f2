"""One variant of tests/test_gpu_linesearch.py: python linesearch_worker.py <kind> <out.npz>, with MUSE_HIP_LIB pointing at the
-DMUSE_HZTEST build of that placement.  Solves the non-quadratic test objective from several starts / tolerances through
muse_zhat_at_theta and saves the MAPs and solver infos."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import museinference_jl_amd as M

kind, out = sys.argv[1], sys.argv[2]
N = {"resident": 10000, "streaming": 7001, "cluster": 70000}[kind]
res = {}
prob = M.HipMuseProblem(None, model="noise", ntheta=1, N=N)
if kind == "streaming":
    prob.set_placement(0)
assert prob.placement_info()["resident"] == (kind == "resident") and (prob.placement_info()["workgroups_per_element"] > 1) == (kind == "cluster")
rng = np.random.default_rng(11)
case = 0
for theta in (-2.0, 0.0, 1.5):
    for scale, start in ((3.0, "zero"), (1.0, "far")):
        for atol in (1e-2, 1e-7):
            x = rng.standard_normal(N) * scale
            z0 = np.zeros(N) if start == "zero" else 4.0 * rng.standard_normal(N)
            z, info = prob.zhat_at_theta(x, z0, [theta], atol)
            res[f"x{case}"], res[f"z0{case}"], res[f"z{case}"] = x, z0, z
            res[f"info{case}"] = np.array([info["iterations"], info["f_calls"], info["status"]])
            res[f"par{case}"] = np.array([theta, atol, info["f_min"], info["gnorm"]])
            case += 1
res["ncases"] = np.array(case)
prob.close()
np.savez(out, **res)
