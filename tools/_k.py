import sys, numpy as np
sys.path.insert(0, '/root/repo')
import museinference_jl_amd as M
nth = int(sys.argv[1])
prob = M.HipMuseProblem(None, model="funnel", ntheta=nth, N=10000)
prob.set_normals_cache(False)
for _ in range(4):
    prob.map_and_score_batch(1, 0, 512, [1.0] * nth, atol=1e-2, z0_mode=0)
prob.close()
