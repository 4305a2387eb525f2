"""Run-to-run determinism stress: the same batch many times, every output compared bitwise with the first."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, museinference_jl_amd as M
cases = [("funnel", 70001, 2, [0.3, 0.1], -1, 9, 1e-3), ("funnel", 70001, 1, [0.3], -1, 9, 1e-3), ("funnel", 70002, 2, [0.3, 0.1], -1, 9, 1e-3), ("noise", 131072, 1, [-0.4], -1, 9, 1e-3), ("funnel", 131072, 2, [0.3, 0.1], -1, 9, 1e-3),
         ("funnel", 10000, 4, [1.0, 0.5, -0.5, 2.0], 0, 12, 1e-3), ("funnel", 10000, 4, [1.0, 0.5, -0.5, 2.0], 1, 12, 1e-3),
         ("funnel", 10000, 1, [1.0], 1, 40, 1e-3), ("funnel", 10000, 1, [1.0], 0, 40, 1e-3),
         ("smooth", 66001, 2, [1.0, 2.5], -1, 7, 1e-2), ("noise", 131072, 1, [-0.4], -1, 7, 1e-2),
         ("smooth", 20000, 8, [1.0, 2.0, 3.0, 0.5, 0.0, -1.0, 1.5, 2.5], -1, 20, 1e-2), ("funnel", 1000, 8, [0.1] * 8, -1, 40, 1e-3)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
for model, N, nth, th, pl, n, atol in cases:
    prob = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
    if pl >= 0:
        prob.set_placement(pl)
    ref = None
    bad = 0
    for r in range(reps):
        g, info = prob.map_and_score_batch(7, 0, n, th, atol=atol, z0_mode=0)
        z = prob.get_zhat(0, n)
        cur = (g.copy(), info.copy(), z)
        if ref is None:
            ref = cur
        elif not (np.array_equal(cur[0], ref[0]) and np.array_equal(cur[1], ref[1]) and np.array_equal(cur[2], ref[2])):
            bad += 1
            if bad == 1:
                dg = np.argwhere(cur[0] != ref[0])
                dz = np.argwhere(cur[2] != ref[2])
                print("   first mismatch: g at", dg[:6].tolist(), "iters", cur[1]["iterations"].tolist(), "vs", ref[1]["iterations"].tolist(), "z mismatches", len(dz), dz[:5].tolist(), "fmin eq", np.array_equal(cur[1]["f_min"], ref[1]["f_min"]))
    print(f"{model:7s} N={N:7d} ntheta={nth} placement={pl:2d}: {bad}/{reps - 1} runs differ from the first")
    prob.close()
