"""Randomized parity of a user-supplied model (museinference.jl_amd/models/cubic.h: a non-quadratic MAP objective, 5-100
L-BFGS iterations with real line searches and a wrapping history) against the oracle's build of the same header: random
N / ntheta / theta / seed / atol / start mode / placement / element split / maps per launch.  Classes of outcome:
  same path   identical iteration, evaluation counts and status -> scores to 1e-6 (relative to the larger of |score| and N, the
              size of its terms) and MAPs to 1e-9 ("tight": what the built-in, quadratic models get) or, after tens of
              iterations of a non-quadratic objective along which the two summation orders drift apart, to 0.1 atol ("drift")
  off path    a decision of the line search fell the other way (tree-ordered against sequential sums over many iterations): the
              counts differ -- or, rarely, coincide while the iterates do not -> both converged, MAPs agree to 2 atol
  MISMATCH    anything else (printed).
Usage (GPU box): python tools/fuzz_user_model.py [seconds] [seed] [cubic|pair|gen|pairgen]
`gen` / `pairgen`: the cubic model / the two-parameter model as headers GENERATED from their terms (ElementwiseModel.from_expressions /
from_pair_expressions) instead of the hand-written ones.
`pair`: the two-parameter family's shipped member (models/normal_mean_var.h: ntheta = 2 K in {2, 4, 6, 8}, location parameters in
[-1.5, 1.5], no implicit-differentiation H; its objective is quadratic, so every case is expected on the same path) -- with the native
loops on every third case: muse_run_device (the loop kernel, whose step calls the header's muse_model_coefs on the device) against
muse_run, bit for bit."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import museinference_jl_amd as M
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
MODE = sys.argv[3] if len(sys.argv) > 3 else "cubic"
PAIR = MODE in ("pair", "pairgen")
if MODE in ("gen", "pairgen"):   # the same two models GENERATED from their terms (museinference_jl_amd.symbolic; tests/test_symbolic_model.py)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import test_symbolic_model as TS
    model = TS.generated_nmv(M) if PAIR else TS.generated_cubic(M)
    NAME = model.library_name
else:
    NAME = "normal_mean_var" if PAIR else "cubic"
    model = M.ElementwiseModel.packaged(NAME)
os.environ.setdefault("MUSE_DEBUG_LOOP_ANY_NTHETA", "1")
t0 = time.time()
nloop = nloop_bad = 0
ncase = nsame = ntight = noff = noff_equal = nbad = nimp = nimp_bad = 0
maxit = 0
with O.user_model(model.header, NAME):
    while time.time() - t0 < budget:
        N = int(rng.choice([int(rng.integers(5, 600)), int(rng.integers(600, 4200)), int(rng.integers(4000, 10100)),
                            int(rng.integers(10000, 30000)), int(rng.integers(65000, 80000))]))
        nth = min(int(rng.choice([1, 2, 3, 4, 8, 8, int(rng.integers(9, 65))])), N)   # (9..64: the big tier)
        theta = rng.uniform(-2.0, 0.8, size=nth)
        if PAIR:
            nth = int(rng.choice([2, 2, 4, 6, 8]))
            N = max(N, nth)
            theta = np.concatenate([rng.uniform(-1.5, 1.5, size=nth // 2), rng.uniform(-2.0, 1.5, size=nth // 2)])
        atol = float(rng.choice([1e-2, 1e-4, 1e-6]))
        z0 = int(rng.choice([0, 1]))
        n = 4 if N < 20000 else 2
        seed, s0 = int(rng.integers(1, 2**40)), int(rng.integers(0, 5000))
        prob = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
        placement = int(rng.choice([-1, -1, 0]))
        if placement >= 0:
            prob.set_placement(placement)
        split = int(rng.choice([0, 0, 2, 4, 8]))
        if split and N > 512:
            prob.set_element_split(split)
        else:
            split = 0
        nmaps = 1 if nth > 8 else int(rng.choice([1, 1, 2, 3]))
        if nmaps > 1:
            thetas = np.vstack([theta] + [theta + rng.uniform(-0.3, 0.3, size=nth) for _ in range(nmaps - 1)])
            tot = prob.map_and_score_multi_async(seed, s0, s0 + n, thetas, atol=atol, z0_mode=z0, result_area=2)
            g, info = prob.batch_wait(tot, 2)
            g, info = g[:n], info[:n]
        else:
            g, info = prob.map_and_score_batch(seed, s0, s0 + n, theta, atol=atol, z0_mode=z0)
        zh = prob.get_zhat(0, n)
        # every fourth case: the implicit-differentiation H of one simulation (the header's second derivatives) against the checker's
        imp = None
        if not PAIR and rng.random() < 0.25 and N >= 20:
            Hi, its = prob.implicit_H_batch(seed, s0, s0 + 1, theta, atol=1e-1, cg_maxiter=100)
            imp = (Hi[0], its[0])
        prob.close()
        if PAIR and ncase % 3 == 0 and N <= 10000 and not split and placement < 0:
            # the native loops on the same shape: ONE persistent launch for all iterations against one launch per iteration
            xd = np.cos(0.37 * np.arange(N)) * 1.3 + 0.2
            lp = M.HipMuseProblem(xd, model=model, ntheta=nth, prior=M.GaussianPrior(0.0, 3.0))
            kw = dict(nsims=int(rng.integers(2, 90)), maxsteps=int(rng.integers(1, 7)), theta_rtol=0.0, atol=atol, alpha=float(rng.uniform(0.4, 1.0)))
            th0 = theta * 0.5
            try:
                a = lp.run_muse(seed, th0, device_loop=False, **kw)
                b = lp.run_muse(seed, th0, device_loop=True, **kw)
                ok = a[0] == b[0] and np.array_equal(a[1], b[1], equal_nan=True) and np.array_equal(a[2][:, :-1], b[2][:, :-1], equal_nan=True) \
                    and np.array_equal(a[3], b[3], equal_nan=True) and a[4].tobytes() == b[4].tobytes()
            except M.MuseError as e:
                ok = "singular" in str(e)
            lp.close()
            nloop += 1
            if not ok:
                nloop_bad += 1
                print("LOOP MISMATCH", "N", N, "nth", nth, theta.tolist(), kw, seed, flush=True)
        if imp is not None:
            Ho, io_cg = O.implicit_H("user", N, seed, s0, theta, atol=1e-1, cg_maxiter=100)
            nimp += 1
            scale = max(np.abs(Ho).max(), 1.0)
            if not (np.all(np.abs(imp[1] - io_cg) <= 2) and np.all(np.abs(imp[0] - Ho) <= 1e-6 * scale)):
                nimp_bad += 1
                print("IMPLICIT MISMATCH", "N", N, "nth", nth, "placement", placement, "split", split, theta.tolist(), seed, s0, imp[1], io_cg,
                      float(np.abs(imp[0] - Ho).max() / scale), flush=True)
        go, zo, io = O.map_and_score_batch("user", N, seed, s0, s0 + n, theta, atol=atol, z0_mode=z0, nthreads=8)
        ncase += 1
        maxit = max(maxit, int(io["iterations"].max()))
        same = (np.array_equal(info["iterations"], io["iterations"]) and np.array_equal(info["f_calls"], io["f_calls"])
                and np.array_equal(info["status"], io["status"]))
        dz = float(np.abs(zh - zo).max())
        if same and dz <= max(1e-9, 0.1 * atol) and np.all(np.abs(g - go) <= 1e-6 * np.maximum(np.abs(go), N)):
            nsame += 1
            ntight += dz <= 1e-9
        elif info["status"].max() <= 2 and io["status"].max() <= 2 and dz <= 2 * atol:
            noff += 1
            noff_equal += bool(same)
        else:
            nbad += 1
            print("MISMATCH", "N", N, "nth", nth, "placement", placement, "split", split, "nmaps", nmaps, theta.tolist(), atol, z0, seed, s0,
                  info["iterations"], io["iterations"], info["f_calls"], io["f_calls"], info["status"], io["status"], dz,
                  float(np.abs(g - go).max()), flush=True)
print(f"{ncase} cases: {nsame} same path ({ntight} of them with MAPs to 1e-9), {noff} off path (converged, MAPs within 2 atol; {noff_equal} of them with equal counts), {nbad} MISMATCHES; longest solve {maxit} iterations; "
      f"implicit-differentiation H: {nimp} cases, {nimp_bad} beyond 1e-6 or two CG iterations; "
      f"native loops (device against host): {nloop} cases, {nloop_bad} mismatches; {time.time() - t0:.0f} s")
