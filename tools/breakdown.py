import sys, numpy as np, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import museinference_jl_amd as M
lib = M.load_library()
N=10000
x = np.random.default_rng(0).normal(size=N)*1.5
for name, flags in [("full",0),("sample+score only",1),("no sampling (x=data), solve",2),("neither",3)]:
    prob = M.HipMuseProblem(x, model="funnel", ntheta=1)
    lib.muse_debug_flags(prob._ctx, flags)
    for _ in range(5): prob.map_and_score_batch(0,0,512,[1.0])
    prob.profile_begin(64)
    for _ in range(50): prob.map_and_score_batch(0,0,512,[1.0])
    ms = prob.profile_end()
    print(f"{name:35s} kernel mean {ms.mean()*1e3:8.1f} us  min {ms.min()*1e3:8.1f} us")
    prob.close()
