import os, sys
os.environ.setdefault("MUSE_DEBUG_LOOP_ANY_NTHETA", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import museinference_jl_amd as M
import test_symbolic_model as TS
cases = [(8891, 8, [0.6481675193853551, 0.44236596767085645, -0.5224277414192845, -0.8274333586054914, 1.1418244157068949, -0.6334131468643509, 0.9279280847931961, -1.3487574220187644], dict(nsims=51, maxsteps=1, theta_rtol=0.0, atol=0.0001, alpha=0.974401881686763), 880591506137),
         (6577, 6, [0.0490589968949835, -1.42091628441271, 0.08532281727109403, -0.9011423199001489, -1.2948087055703406, 0.6377306930679851], dict(nsims=20, maxsteps=6, theta_rtol=0.0, atol=0.01, alpha=0.5957749455803162), 697790958300)]
for name, model in (("generated", TS.generated_nmv(M)), ("hand-written", M.ElementwiseModel.packaged("normal_mean_var"))):
    for N, nth, th0, kw, seed in cases:
        x = np.sin(0.3 * np.arange(N)) + 0.4 + 0.8 * np.cos(1.7 * np.arange(N))
        prob = M.HipMuseProblem(x, model=model, ntheta=nth, prior=M.GaussianPrior(0.0, 3.0))
        a = prob.run_muse(seed, th0, device_loop=False, **kw)
        b = prob.run_muse(seed, th0, device_loop=True, **kw)
        b2 = prob.run_muse(seed, th0, device_loop=True, **kw)
        a2 = prob.run_muse(seed, th0, device_loop=False, **kw)
        print(name, N, nth, "iters", a[0], b[0], "theta eq", np.array_equal(a[1], b[1]), "hist eq", np.array_equal(a[2][:, :-1], b[2][:, :-1], equal_nan=True),
              "gs eq", np.array_equal(a[3], b[3]), "info eq", a[4].tobytes() == b[4].tobytes(), "| dev repeat eq", np.array_equal(b[3], b2[3]) and np.array_equal(b[1], b2[1]),
              "host repeat eq", np.array_equal(a[3], a2[3]))
        if not np.array_equal(a[3], b[3]):
            d = np.argwhere(a[3] != b[3])
            print("  gs differ at", d[:6].tolist(), "of shape", a[3].shape, "max rel", np.max(np.abs(a[3] - b[3]) / np.maximum(np.abs(a[3]), 1e-300)))
            print("  host", a[3][tuple(d[0])], "dev", b[3][tuple(d[0])])
        if not np.array_equal(a[2][:, :-1], b[2][:, :-1], equal_nan=True):
            d = np.argwhere(~((a[2][:, :-1] == b[2][:, :-1]) | (np.isnan(a[2][:, :-1]) & np.isnan(b[2][:, :-1]))))
            print("  hist differs at", d[:8].tolist(), a[2].shape)
        if a[4].tobytes() != b[4].tobytes():
            ia, ib = a[4], b[4]
            print("  shape", ia.shape, ia.dtype.names)
            for f in ia.dtype.names:
                print("   ", f, "host", ia[f].ravel()[:8].tolist(), "dev", ib[f].ravel()[:8].tolist())
        prob.close()
