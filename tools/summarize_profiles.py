"""Condense the rocprofv3 outputs of tools/profile.sh (gpurun_out/prof/) into profiles/: per workload the
kernel-stats CSV of the --kernel-trace --stats run, and one summary row with the solver kernel's average
duration and the PMC counters of the separate --pmc passes (averages per solver launch).
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE under-reports by 2x on gfx950
(MI355X_MICROARCH.md, HBM / rocprofv3 section); both counters are in KiB."""
import csv, glob, importlib.util, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
_bench = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_bench)
# the kernel sources the passes ran on: tools/profile.sh records it on the GPU box (the same tree), else this tree's
_shafile = os.path.join(ROOT, "gpurun_out", "prof", "csrc_sha16.txt")
CSRC_SHA = open(_shafile).read().strip() if os.path.exists(_shafile) else _bench.csrc_fingerprint()
PROF = os.path.join(ROOT, "gpurun_out", "prof")
OUT = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
workloads = sorted({os.path.basename(d)[len("trace_"):] for d in glob.glob(os.path.join(PROF, "trace_*"))})


def counters(dirname):
    """{counter: mean value per solver-kernel dispatch} from a --pmc pass."""
    acc = {}
    for f in glob.glob(os.path.join(PROF, dirname, "*", "*_counter_collection.csv")):
        per_dispatch = {}
        for row in csv.DictReader(open(f)):
            if "map_score_kernel" not in row["Kernel_Name"]:
                continue
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])
        for (_, name), v in per_dispatch.items():
            acc.setdefault(name, []).append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


rows = []
for w in workloads:
    stats = glob.glob(os.path.join(PROF, "trace_" + w, "*", "*_kernel_stats.csv"))
    if not stats:
        continue
    if len(stats) > 1:  # gpurun merges a pass into gpurun_out/: files of an earlier pass must not be mixed in
        sys.exit(f"gpurun_out/prof/trace_{w} holds {len(stats)} passes: rm -rf gpurun_out/prof and run tools/profile.sh again")
    shutil.copy(stats[0], os.path.join(OUT, f"{tag}_{w}_kernel_stats.csv"))
    k = [r for r in csv.DictReader(open(stats[0])) if "map_score_kernel" in r["Name"]][0]
    c = {}
    for d in ("pmc_fetch_", "pmc_write_", "pmc_sq_"):
        c.update(counters(d + w))
    fetch, write = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
    hbm_mb = (2 * fetch + write) * 1024 / 1e6 if fetch is not None and write is not None else ""
    rows.append({"workload": w, "csrc_sha16": CSRC_SHA, "kernel": k["Name"][:120], "calls": k["Calls"], "avg_us": float(k["AverageNs"]) / 1e3,
                 "min_us": float(k["MinNs"]) / 1e3, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
                 "hbm_traffic_MB": hbm_mb, **{n: c.get(n, "") for n in
                 ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU",
                  "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE")}})
with open(os.path.join(OUT, f"{tag}_summary.csv"), "w", newline="") as f:
    wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    wr.writeheader()
    wr.writerows(rows)
bj = os.path.join(PROF, "bench_full.json")
if os.path.exists(bj):
    lines = [l for l in open(bj) if l.startswith("{")]
    if lines:
        open(os.path.join(OUT, f"{tag}_bench_funnel_1e4.json"), "w").write(lines[-1])
for r in rows:
    print(r["workload"], f"{r['avg_us']:.1f} us", "HBM MB/launch", r["hbm_traffic_MB"])
