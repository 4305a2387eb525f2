"""Diagnostic: in-kernel stamp breakdown of the LAST iteration kernel of a native muse! loop (muse_run: nsims + 1 warm
elements, normals from the cache) from the -DMUSE_STAMPS build (never timed):
    python tools/stamps_run.py [N] [ntheta] [nsims] [iterations]
Prints the per-phase shader-cycle medians of a problem, the rounds the launch ran in and where the data element sat."""
import os, sys, numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import museinference_jl_amd as M
from museinference_jl_amd import build as B
B.LIB_PATH = B.LIB_PATH.replace("libmuse_hip.so", os.environ.get("MUSE_STAMPS_LIB", "libmuse_hip_stamps.so"))
lib = M.load_library()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
NTH = int(sys.argv[2]) if len(sys.argv) > 2 else 1
S = int(sys.argv[3]) if len(sys.argv) > 3 else 512
ITERS = int(sys.argv[4]) if len(sys.argv) > 4 else 4
DEV = len(sys.argv) > 5 and sys.argv[5] == "dev"    # the device-resident loop (muse_run_device) instead of the host loop
xdata, _ = M.HipMuseProblem(None, model="funnel", ntheta=NTH, N=N).sample_x_z(M.SimRng(0, M.DATA_SIM), [0.0] * NTH)
prob = M.HipMuseProblem(xdata, model="funnel", ntheta=NTH, prior=M.GaussianPrior(0.0, 3.0))
n = S + 1
if os.environ.get("MUSE_LOOP_DEBUG"):   # bit 2 (4): the loop kernel does not prefetch
    lib.muse_debug_flags(prob._ctx, int(os.environ["MUSE_LOOP_DEBUG"]))
for _ in range(2):
    prob.run_muse(0, [1.0] * NTH, nsims=S, maxsteps=ITERS, theta_rtol=1e-12, atol=1e-2, alpha=0.7, device_loop=DEV)
lib.muse_debug_stamps(prob._ctx, C.c_int64(n + 8), None)
nit, theta, hist, gs, info = prob.run_muse(0, [1.0] * NTH, nsims=S, maxsteps=ITERS, theta_rtol=1e-12, atol=1e-2, alpha=0.7, device_loop=DEV)
out_all = np.zeros((n + 8, 16), dtype=np.uint64)
lib.muse_debug_stamps(prob._ctx, C.c_int64(n + 8), out_all.ctypes.data_as(C.c_void_p))
out = out_all[:n]
o = out.astype(np.int64)
if DEV:   # the loop kernel's own stamps of its last iteration (100 MHz clock -> us), relative to worker 0's iteration start
    lp = out_all[n:n + 3].astype(np.int64)
    t0 = lp[0, 0]
    for nm, r in (("workgroup 0 (an element more)", lp[0]), ("a worker in the middle", lp[1]), ("the stepper", lp[2])):
        print(f"  loop stamps, {nm:28s}: " + "  ".join(f"{k}:{(r[k] - t0) / 100.0:7.2f}" for k in range(6) if r[k] > 0),
              " [0 iteration start, 1 first problem done, 2 problems done, 3 prefetch issued, 4 theta received / sweep complete, 5 theta published]")
st = o[:, :8]
print(f"N={N} ntheta={NTH} nsims={S}: iteration {nit} of {'muse_run_device' if DEV else 'muse_run'}; iterations/f_calls of its solves: "
      f"{info[-1]['iterations'].mean():.2f} / {info[-1]['f_calls'].mean():.2f}")
d = np.diff(st, axis=1)
names = ["load x / normals / z0", "eval0 (+trial)", "twoloop (s=-g)", "linesearch", "update (+score, zhat)", "(loop exit)", "finish"]
tot = np.median(st[:, 7] - st[:, 0])
print("median shader cycles per phase (thread 0 of the workgroup):")
for k, nm in enumerate(names):
    print(f"  {nm:24s} {np.median(d[:, k]):9.0f}  ({100 * np.median(d[:, k]) / tot:5.1f} %)")
print(f"  total per problem        {tot:9.0f}")
print("  line search detail: pre-logic", np.median(o[:, 10] - st[:, 3]), " eval1", np.median(o[:, 11] - o[:, 10]), " logic1",
      np.median(o[:, 12] - o[:, 11]), " eval2", np.median(o[:, 13] - o[:, 12]), " post-logic", np.median(st[:, 4] - o[:, 13]))
print("  final update pass: element loop", np.median(o[:, 14] - st[:, 4]), " reduction", np.median(o[:, 15] - o[:, 14]), " rest",
      np.median(st[:, 5] - o[:, 15]))
t0 = st[:, 0].min()
start, end = st[:, 0] - t0, st[:, 7] - t0
order = np.argsort(start)
print("problem start offsets (cycles), sorted: ", start[order][[0, 1, n // 4, n // 2 - 2, n // 2, 3 * n // 4, n - 2, n - 1]])
print("problem end offsets (cycles), sorted:   ", np.sort(end)[[0, 1, n // 4, n // 2 - 2, n // 2, 3 * n // 4, n - 2, n - 1]])
print(f"data element (problem 0): start {start[0]}, end {end[0]}, length {end[0] - start[0]}")
late = order[-3:]
print("the three problems that started last:", [(int(p), int(start[p]), int(end[p])) for p in late])
print(f"launch span (first start -> last end): {end.max()} cycles; without the last-started problem: {np.sort(end)[-2]}")
ends = np.sort(st[:, 7]); gaps = []
for s0 in st[:, 0]:
    k = np.searchsorted(ends, s0) - 1
    if k >= 0 and 0 < s0 - ends[k] < 20000: gaps.append(s0 - ends[k])
if gaps: print("between two problems of a workgroup (end stamp -> next start stamp): median", np.median(gaps), "cycles over", len(gaps))
if DEV:
    # (the workgroups that own elements on a 256-CU part: a worker per element and a stepper beside them, or -- more elements than
    #  that -- all 256 with the stepper, the last one, among them; MUSE_LOOP_DEBUG=128: the stepper never solves)
    dedicated = n <= 255 or (int(os.environ.get("MUSE_LOOP_DEBUG", "0")) & 128) or os.environ.get("MUSE_DEBUG_LOOP_DEDICATED_STEPPER")
    nw = min(255, n) if dedicated else 256
    for wk in (0, 1, 100) + (() if dedicated else (255,)):
        ps = [q for q in range(wk, n, nw)]
        print(f"  worker {wk}: problems {ps}: " + " | ".join(
            f"p{q}: begin {st[q,1]-st[q,0]} solve {st[q,6]-st[q,1]} total {st[q,7]-st[q,0]} (start +{st[q,0]-st[ps[0],0]})" for q in ps))
print("iteration wall times from the history records (us):", np.round(1e6 * hist[:, -1], 1))
