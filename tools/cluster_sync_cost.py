"""Diagnostic: kernel time with the cluster waits disabled (results are garbage; timing only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, museinference_jl_amd as M
lib = M.load_library()
for model, N, nth, th, n in [("smooth", 100000, 8, [1.0] * 8, 128), ("noise", 1000000, 1, [0.5], 128)]:
    for flags in (0, 4):
        prob = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
        lib.muse_debug_flags(prob._ctx, flags)
        for _ in range(2):
            g, info = prob.map_and_score_batch(0, 0, n, th)
        prob.profile_begin(8)
        for _ in range(3):
            g, info = prob.map_and_score_batch(0, 0, n, th)
        ms = prob.profile_end()
        print(f"{model} N={N} debug={flags}: kernel {ms.mean():.3f} ms; iterations mean {info['iterations'].mean():.2f} f_calls {info['f_calls'].mean():.2f}")
        prob.close()
