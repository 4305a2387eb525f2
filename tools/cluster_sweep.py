"""Diagnostic: solver kernel time vs cluster size for the large-N workloads (MUSE_DEBUG_CLUSTER_SIZE)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np, museinference_jl_amd as M
model, N, nth, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
th = [0.5] if model == "noise" else [1.0] * nth
prob = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
for _ in range(2):
    g, info = prob.map_and_score_batch(0, 0, n, th)
prob.profile_begin(8)
for _ in range(3):
    g, info = prob.map_and_score_batch(0, 0, n, th)
ms = prob.profile_end()
E = info["f_calls"].astype(np.int64); K = info["iterations"].astype(np.int64); H = info["hist_words"].astype(np.int64)
words = (1 + 5 * E + 4 * H + 4 * K + 2).sum()
print(f"{model} N={N} cs={os.environ.get('MUSE_DEBUG_CLUSTER_SIZE','auto')}: kernel {ms.mean():.3f} ms; K mean {K.mean():.2f} max {K.max()} E mean {E.mean():.2f} hist {H.mean():.1f}; alg {8*N*words/1e9:.2f} GB -> {8*N*words/ms.mean()/1e9:.0f} TB/s-milli", flush=True)
''' % ROOT
for model, N, nth, n in [("smooth", 100000, 8, 128), ("noise", 1000000, 1, 128), ("funnel", 100000, 8, 128), ("smooth", 20000, 2, 512)]:
    for cs in sys.argv[1:] or ["auto"]:
        env = dict(os.environ)
        if cs != "auto":
            env["MUSE_DEBUG_CLUSTER_SIZE"] = cs
        r = subprocess.run([sys.executable, "-c", code, model, str(N), str(nth), str(n)], env=env, capture_output=True, text=True)
        print((r.stdout.strip() or r.stderr.strip()[-400:]))
