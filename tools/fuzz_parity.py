"""Long randomized parity run against the CPU oracle (development aid; the committed tests hold a fixed subset):
random model / N / ntheta / theta / seed / atol / start mode / batch size (N >= 65 536: sometimes more problems than clusters); compares iteration and evaluation counts, status,
scores (rtol 1e-9) and zhat (atol 1e-9).  Usage: python tools/fuzz_parity.py [seconds] [seed] [big]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, museinference_jl_amd as M
from oracle import oracle as O
O.build()
BIG = "big" in sys.argv[3:]     # python tools/fuzz_parity.py seconds seed big: every case of a non-noise model has 9..64 components
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); ncase = nbad = 0
while time.time() - t0 < budget:
    model = str(rng.choice(["funnel", "noise", "smooth"]))
    N = int(rng.choice([int(rng.integers(5, 600)), int(rng.integers(600, 4200)), int(rng.integers(4000, 10100)),
                        int(rng.integers(10000, 40000)), int(rng.integers(65000, 90000))]))
    nth = 1 if model == "noise" else int(rng.choice([1, 2, 3, 4, 8]))
    if BIG and model != "noise":
        nth = int(rng.integers(9, 65))   # the big tier (ntheta > MUSE_MAX_THETA: streaming placements, one map per launch)
    nth = min(nth, N)
    theta = rng.uniform(-1.5, 2.5, size=nth)
    atol = float(rng.choice([1e-2, 1e-4, 1e-6]))
    z0 = int(rng.choice([0, 1]))
    n = 4 if N < 20000 else 2
    if N >= 65536 and model != "smooth" and rng.random() < 0.3:
        n = int(rng.integers(66, 90))  # more problems than streaming clusters: the background generator draws the later ones
    seed, s0 = int(rng.integers(1, 2**40)), int(rng.integers(0, 5000))
    prob = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
    split = int(rng.choice([0, 0, 2, 4, 8]))  # element split: resident clusters (N <= 10^4, elementwise) or streaming ones
    if split and N > 512:
        prob.set_element_split(split)
    else:
        split = 0
    nmaps = 1 if nth > 8 else int(rng.choice([1, 1, 1, 2, 3]))  # several maps in one launch: map 0 is compared, the others carry other thetas
    if nmaps > 1:
        thetas = np.vstack([theta] + [rng.uniform(-1.5, 2.5, size=nth) for _ in range(nmaps - 1)])
        tot = prob.map_and_score_multi_async(seed, s0, s0 + n, thetas, atol=atol, z0_mode=z0, result_area=2)
        g, info = prob.batch_wait(tot, 2)
        g, info = g[:n], info[:n]
    else:
        g, info = prob.map_and_score_batch(seed, s0, s0 + n, theta, atol=atol, z0_mode=z0)
    zh = prob.get_zhat(0, n)
    prob.close()
    go, zo, io = O.map_and_score_batch(model, N, seed, s0, s0 + n, theta, atol=atol, z0_mode=z0, nthreads=8)
    ok = (np.array_equal(info["iterations"], io["iterations"]) and np.array_equal(info["f_calls"], io["f_calls"])
          and np.array_equal(info["status"], io["status"]) and np.allclose(g, go, rtol=1e-9, atol=1e-9)
          and np.allclose(zh, zo, rtol=0, atol=1e-9))
    ncase += 1
    if not ok:
        nbad += 1
        print("MISMATCH", "nmaps", nmaps, "split", split, model, N, nth, theta.tolist(), atol, z0, seed, s0, info["iterations"], io["iterations"],
              info["f_calls"], io["f_calls"], info["status"], io["status"], np.abs(g - go).max(), flush=True)
print(f"{ncase} cases, {nbad} mismatches in {time.time() - t0:.0f} s")
