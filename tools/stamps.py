"""Diagnostic: per-phase shader-clock shares of one batch from the -DMUSE_STAMPS build (never timed)."""
import sys, numpy as np, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import museinference_jl_amd as M
from museinference_jl_amd import build as B, _capi
B.LIB_PATH = B.LIB_PATH.replace("libmuse_hip.so", __import__("os").environ.get("MUSE_STAMPS_LIB", "libmuse_hip_stamps.so"))
lib = M.load_library()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
NTH = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 512
prob = M.HipMuseProblem(None, model="funnel", ntheta=NTH, N=N)
if len(sys.argv) > 4: prob.set_element_split(int(sys.argv[4]))
for _ in range(3): prob.map_and_score_batch(0,0,n,[1.0]*NTH)
lib.muse_debug_stamps(prob._ctx, C.c_int64(n), None)
prob.map_and_score_batch(0,0,n,[1.0]*NTH)
out = np.zeros((n,16), dtype=np.uint64)
lib.muse_debug_stamps(prob._ctx, C.c_int64(n), out.ctypes.data_as(C.c_void_p))
st = out[:, :8].astype(np.int64)
d = np.diff(st, axis=1)
names = ["sample/load", "eval0", "twoloop(s=-g)", "linesearch", "update", "(loop exit)", "score+zhat"]
print("median shader cycles per phase (wave 0):")
for k, nm in enumerate(names): print(f"  {nm:16s} {np.median(d[:,k]):10.0f}   ({np.median(d[:,k])/np.median(st[:,7]-st[:,0])*100:5.1f} %)")
ent = o0 = out.astype(np.int64)
first = (ent[:, 8] > 0) & (ent[:, 8] < st[:, 0]) & (st[:, 0] - ent[:, 8] < 10**7)
if first.any(): print("  kernel entry -> first stamp of the workgroup's first problem: median", np.median(st[first, 0] - ent[first, 8]), "cycles over", int(first.sum()), "workgroups")
last = (ent[:, 9] > st[:, 7]) & (ent[:, 9] - st[:, 7] < 10**7)
if last.any(): print("  last stamp of the last problem -> end of the kernel's loop: median", np.median(ent[last, 9] - st[last, 7]), "cycles over", int(last.sum()))
print("  sample detail: start->sampler loop end", np.median(out[:,8].astype(np.int64)-st[:,0]), " ->z init end", np.median(out[:,9].astype(np.int64)-out[:,8].astype(np.int64)), " ->barrier end", np.median(st[:,1]-out[:,9].astype(np.int64)))
o=out.astype(np.int64)
print("  line search detail: pre-logic", np.median(o[:,10]-st[:,3]), " eval1", np.median(o[:,11]-o[:,10]), " logic1", np.median(o[:,12]-o[:,11]), " eval2", np.median(o[:,13]-o[:,12]), " post-logic", np.median(st[:,4]-o[:,13]))
print("  final update pass: element loop", np.median(o[:,14]-st[:,4]), " reduction", np.median(o[:,15]-o[:,14]), " rest", np.median(st[:,5]-o[:,15]))
print("total per problem", np.median(st[:,7]-st[:,0]), "cycles; first start -> last end:", (st[:,7].max()-st[:,0].min()), "cycles")
# a workgroup's next problem starts right after its previous one ended (same CU, same counter): the smallest positive
# start - end distance of each problem to any other is the per-problem prologue/epilogue outside the stamped span
ends = np.sort(st[:, 7]); gaps = []
for s0 in st[:, 0]:
    k = np.searchsorted(ends, s0) - 1
    if k >= 0 and 0 < s0 - ends[k] < 20000: gaps.append(s0 - ends[k])
if gaps: print("between two problems of a workgroup (end stamp -> next start stamp): median", np.median(gaps), "cycles over", len(gaps), "pairs")
order = np.argsort(st[:,0]); 
print("start offsets of problems (cycles, sorted) sample:", (st[order,0]-st[:,0].min())[[0,1,n//5,n//2-1,n//2,n-1]])
