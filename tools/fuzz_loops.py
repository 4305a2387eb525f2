"""Randomized checks of the round-3 and round-4 paths on a GPU (development aid; the committed tests hold fixed cases):
  * muse_run_device (round 4: the loop kernel) against muse_run: the same number of iterations and the same bits (history,
    scores, infos, theta), also with more elements than workers;
  * the cross-call normals cache: sequences of plain maps on a caching context against a non-caching one (bitwise);
  * several maps in one launch against separate launches (bitwise);
  * muse_fd_values_columns against the oracle's per-simulation operators (rtol 1e-7);
  * round 5: muse_run_sharded over a one-rank communicator (persistent launch with the score board in device or in pinned host memory,
    or the host-driven loop) against muse_run (bitwise).
Usage: python tools/fuzz_loops.py [seconds] [seed]"""
import os
os.environ.setdefault("MUSE_DEBUG_LOOP_ANY_NTHETA", "1")   # (the loop kernel whatever ntheta: muse_run_device's default hands ntheta > 1 to the host loop)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import museinference_jl_amd as M
from oracle import oracle as O
from oracle_problem import OracleBatchedProblem
O.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
n = {"loop": 0, "multi": 0, "fd": 0, "lanes": 0, "cache": 0, "sharded": 0}
bad = 0
while time.time() - t0 < budget:
    model = str(rng.choice(["funnel", "noise", "smooth"]))
    N = int(rng.choice([int(rng.integers(8, 600)), int(rng.integers(600, 4200)), int(rng.integers(4000, 10100)), int(rng.integers(10000, 30000))]))
    nth = 1 if model == "noise" else min(N, int(rng.choice([1, 2, 3, 4, 8])))
    seed = int(rng.integers(1, 2**40))
    kind = str(rng.choice(["loop", "loop", "multi", "fd", "lanes", "cache", "sharded"]))
    x = rng.standard_normal(N) * 1.3
    prob = M.HipMuseProblem(x, model=model, ntheta=nth, prior=M.GaussianPrior(0.0, 3.0) if rng.random() < 0.7 else None)
    ok, why = True, ""
    try:
        if kind == "loop":
            nsims = int(rng.integers(2, 200)) if N < 5000 else int(rng.integers(2, 64))
            if rng.random() < 0.25 and N <= 10000:   # more elements than the loop kernel has workers: two and three per worker, the prefetches
                nsims = int(rng.integers(250, 800))
            th0 = rng.uniform(-0.5, 1.5, size=nth)
            kw = dict(nsims=nsims, maxsteps=int(rng.integers(1, 12)), theta_rtol=float(rng.choice([0.0, 1e-2, 1e-1, 1.0])),
                      atol=float(rng.choice([1e-2, 1e-4])), alpha=float(rng.uniform(0.3, 1.0)), z0_warm=bool(rng.random() < 0.2))
            def run(dev):
                if kw["z0_warm"]:   # the same resident MAPs under both loops: those of a cold map at th0
                    prob.map_and_score_batch(seed, 0, nsims, th0, include_data=True, atol=kw["atol"])
                try:
                    return prob.run_muse(seed, th0, device_loop=dev, **kw)
                except M.MuseError as e:
                    return str(e)
            a, b = run(False), run(True)
            if isinstance(a, str) or isinstance(b, str):
                ok = a == b   # the same error from both loops
            else:
                ok = (a[0] == b[0] and np.array_equal(a[1], b[1], equal_nan=True) and np.array_equal(a[2][:, :-1], b[2][:, :-1], equal_nan=True)
                      and np.array_equal(a[3], b[3], equal_nan=True) and a[4].tobytes() == b[4].tobytes())   # (records with NaN fields: bytes)
            why = f"{kw} {a if isinstance(a, str) else a[0]} {b if isinstance(b, str) else b[0]}"
            if not ok and not isinstance(a, str) and not isinstance(b, str) and a[0] == b[0]:   # which array, where
                for name, u, v in (("theta", a[1], b[1]), ("hist", a[2][:, :-1], b[2][:, :-1]), ("gs", a[3], b[3])):
                    w = np.argwhere(~((u == v) | (np.isnan(u) & np.isnan(v))))
                    if len(w):
                        why += f" | {name} differs at {w[:3].tolist()}: {u[tuple(w[0])]!r} vs {v[tuple(w[0])]!r}"
                w = np.argwhere(a[4] != b[4]) if a[4].tobytes() != b[4].tobytes() else []
                if len(w):
                    why += f" | info differs at {w[:3].tolist()}: {a[4][tuple(w[0])]} vs {b[4][tuple(w[0])]}"
                why += f" | th0 {th0.tolist()} prior {prob.prior}"
        elif kind == "sharded":
            # round 5: the sharded loop (muse_run_sharded over a ONE-rank shared-memory communicator) -- a persistent launch whose scores
            # travel through a board in device memory (hipIpc mapping) or in pinned host memory, or the host-driven loop -- against
            # the unsharded host loop: the same bits
            nsims = int(rng.integers(2, 200)) if N < 5000 else int(rng.integers(2, 64))
            if rng.random() < 0.25 and N <= 10000:
                nsims = int(rng.integers(250, 700))
            th0 = rng.uniform(-0.5, 1.5, size=nth)
            kw = dict(nsims=nsims, maxsteps=int(rng.integers(1, 10)), theta_rtol=float(rng.choice([0.0, 1e-2, 1e-1])),
                      atol=float(rng.choice([1e-2, 1e-4])), alpha=float(rng.uniform(0.3, 1.0)))
            board = str(rng.choice(["ipc", "host", "hostloop"]))
            try:
                a = prob.run_muse(seed, th0, device_loop=False, **kw)
            except M.MuseError as e:
                a = str(e)
            shp = M.HipMuseProblem(x, model=model, ntheta=nth, prior=prob.prior)
            shp.comm_init(1, 0, M.HipMuseProblem.comm_unique_id("shm", max(4096, (nsims + 1) * nth)))   # (the host-driven loop gathers a block of nsims + 1 rows)
            shp.debug_flags({"ipc": 0, "host": M.HipMuseProblem.DEBUG_HOST_BOARD, "hostloop": M.HipMuseProblem.DEBUG_SHARDED_HOST_LOOP}[board])
            try:
                b = shp.run_muse_sharded(seed, th0, **kw)
            except M.MuseError as e:
                b = str(e)
            shp.close()
            if isinstance(a, str) or isinstance(b, str):
                ok = isinstance(a, str) and isinstance(b, str) and (("singular" in a) == ("singular" in b))
            else:
                ok = (a[0] == b[0] and np.array_equal(a[1], b[1], equal_nan=True) and np.array_equal(a[2][:, :-1], b[2][:, :-1], equal_nan=True)
                      and np.array_equal(a[3], b[3], equal_nan=True) and a[4].tobytes() == b[4].tobytes())
            why = f"board {board} {kw} {a if isinstance(a, str) else a[0]} {b if isinstance(b, str) else b[0]}"
        elif kind == "cache":
            # the cross-call normals cache (round 4): a sequence of plain maps with repeated, nested and disjoint simulation ranges,
            # cold / true-z / warm starts, on a context that caches against one that does not -- the same bits
            ref = M.HipMuseProblem(x, model=model, ntheta=nth, prior=prob.prior)
            ref.set_normals_cache(False)
            ranges = [(int(a), int(a) + int(c)) for a, c in zip(rng.integers(0, 40, size=4), rng.integers(1, 70, size=4))]
            seq = [ranges[int(rng.integers(0, len(ranges)))] for _ in range(12)]
            if rng.random() < 0.5:
                seq = [seq[0]] * 4 + seq   # the same range several times in a row: store, then loads
            for (lo_, hi_) in seq:
                th = rng.uniform(-1.0, 2.0, size=nth)
                z0 = int(rng.choice([0, 1, 2]))
                incl = bool(rng.random() < 0.5)
                ga, ia = prob.map_and_score_batch(seed, lo_, hi_, th, include_data=incl, atol=1e-3, z0_mode=z0)
                gb, ib = ref.map_and_score_batch(seed, lo_, hi_, th, include_data=incl, atol=1e-3, z0_mode=z0)
                ok = ok and np.array_equal(ga, gb, equal_nan=True) and ia.tobytes() == ib.tobytes()
            ok = ok and np.array_equal(prob.get_zhat(0, 1), ref.get_zhat(0, 1))
            ref.close()
            why = f"ranges {ranges}"
        elif kind == "multi":
            nmaps, nsims, incl = int(rng.integers(2, 9)), int(rng.integers(1, 40)) if N < 20000 else 3, bool(rng.random() < 0.5)
            split = int(rng.choice([0, 0, 2, 4]))
            if split and N > 512:
                prob.set_element_split(split)
            thetas = rng.uniform(-1.0, 2.0, size=(nmaps, nth))
            atol, z0 = float(rng.choice([1e-2, 1e-5])), int(rng.choice([0, 1]))
            tot = prob.map_and_score_multi_async(seed, 3, 3 + nsims, thetas, include_data=incl, atol=atol, z0_mode=z0, result_area=1)
            g, info = prob.batch_wait(tot, 1)
            per = tot // nmaps
            for m in range(nmaps):
                gm, im = prob.map_and_score_batch(seed, 3, 3 + nsims, thetas[m], include_data=incl, atol=atol, z0_mode=z0)
                ok = ok and np.array_equal(g[m * per:(m + 1) * per], gm, equal_nan=True) and np.array_equal(info[m * per:(m + 1) * per], im)
            why = f"nmaps {nmaps} nsims {nsims} data {incl} split {split}"
        elif kind == "lanes":
            # a pipelined sequence of different maps over the result areas with 2-4 lanes against one launch after the other
            nel, nl = (int(rng.integers(1, 60)) if N < 20000 else 3), int(rng.integers(2, 5))
            split = int(rng.choice([0, 0, 2, 4]))
            if split and N > 512:
                prob.set_element_split(split)
            maps = [(int(rng.integers(0, 50)), rng.uniform(-1.0, 2.0, size=nth), int(rng.choice([0, 1]))) for _ in range(7)]
            want = [prob.map_and_score_batch(seed, s0, s0 + nel, th, atol=1e-3, z0_mode=z0) for s0, th, z0 in maps]
            prob.set_concurrency(nl)
            pend, got = [], {}
            for k, (s0, th, z0) in enumerate(maps):
                pend.append((k, prob.map_and_score_batch_async(seed, s0, s0 + nel, th, atol=1e-3, z0_mode=z0, result_area=k % 4)))
                if len(pend) > 3:
                    kk, nn = pend.pop(0)
                    got[kk] = prob.batch_wait(nn, kk % 4)
            for kk, nn in pend:
                got[kk] = prob.batch_wait(nn, kk % 4)
            ok = all(np.array_equal(got[k][0], want[k][0], equal_nan=True) and np.array_equal(got[k][1], want[k][1]) for k in range(len(maps)))
            why = f"lanes {nl} nel {nel} split {split}"
        else:
            if N > 3000:
                prob.close()
                continue
            G, nsims = int(rng.integers(1, 7)), int(rng.integers(1, 4))
            th0 = rng.uniform(-0.3, 1.0, size=nth)
            lo = int(rng.integers(0, nth))
            hi = int(rng.integers(lo + 1, nsims * nth + 1))
            per_unit = bool(rng.random() < 0.5)
            off = rng.uniform(-0.05, 0.05, size=((hi - lo) if per_unit else nth, G))
            fid = int(rng.choice([0, 1]))
            F, info = prob.fd_values_columns(seed, 2, lo, hi, th0, off, per_unit=per_unit, atol=1e-6, fid_mode=fid)
            orc = OracleBatchedProblem(x, model, nth, nthreads=1)
            Fo, io = orc.fd_values_columns(seed, 2, lo, hi, th0, off, per_unit=per_unit, atol=1e-6, fid_mode=fid)
            ok = np.allclose(F, Fo, rtol=1e-7, atol=1e-7 * max(1.0, np.abs(Fo).max())) and np.array_equal(info["status"], io["status"])
            why = f"G {G} cols [{lo},{hi}) per_unit {per_unit} fid {fid} maxdiff {np.abs(F - Fo).max():.2e}"
    except M.MuseError as e:
        ok, why = False, f"MuseError {e}"
    n[kind] += 1
    if not ok:
        bad += 1
        print("MISMATCH", kind, model, N, nth, seed, why, flush=True)
    prob.close()
print(f"{n} cases, {bad} mismatches in {time.time() - t0:.0f} s")
