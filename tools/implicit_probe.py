"""Diagnostic: wall time of the implicit-differentiation H batch against the finite-difference batch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, museinference_jl_amd as M
for model, N, nth, ns in [("funnel", 10000, 4, 64), ("smooth", 100000, 8, 16), ("noise", 1000000, 1, 16)]:
    prob = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
    th = [0.5] if model == "noise" else [1.0] * nth
    for name, fn in (("implicit", lambda: prob.implicit_H_batch(0, 0, ns, th)), ("fd", lambda: prob.fd_jacobian_batch(0, 0, ns, th, [0.05] * nth))):
        fn()
        t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
        extra = f"cg iterations mean {r[1].mean():.1f}" if name == "implicit" else ""
        print(f"{model} N={N} ntheta={nth} nsims={ns}: {name} {dt*1e3:.3f} ms {extra}")
    prob.close()
