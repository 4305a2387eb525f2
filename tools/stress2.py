import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, museinference_jl_amd as M
th = [1.0, 0.5, -0.5, 2.0]
ref = None
with_cluster = len(sys.argv) > 1 and sys.argv[1] == "cluster"
for it in range(25):
    if with_cluster:
        p = M.HipMuseProblem(None, model="smooth", ntheta=2, N=66001)
        p.map_and_score_batch(42, 3, 10, [1.0, 2.5], atol=1e-2)
        p.close()
    res = []
    for placement in (0, 1):
        prob = M.HipMuseProblem(None, model="funnel", ntheta=4, N=10000)
        prob.set_placement(placement)
        g, info = prob.map_and_score_batch(7, 0, 12, th, atol=1e-3, z0_mode=0)
        res.append((g, info["iterations"].copy(), prob.get_zhat(0, 12)))
        prob.close()
    if ref is None:
        ref = res[0]
    for pl in (0, 1):
        if not np.array_equal(res[pl][0], ref[0]):
            bad = np.argwhere(res[pl][0] != ref[0])
            print(f"iter {it} placement {pl}: g differs at {bad.tolist()[:6]}; z equal: {np.array_equal(res[pl][2], ref[2])}; "
                  f"maxdiff {np.abs(res[pl][0]-ref[0]).max():.3e}")
print("done", "with cluster kernel interleaved" if with_cluster else "")
