"""Condense tools/profile_run.sh's output (gpurun_out/prof_run/: rocprofv3 passes of the native muse! loops at configs[1]) into
profiles/<tag>_muse_run_*: the kernel-stats CSVs, one summary row per loop (the iteration kernel's average duration, HBM
bytes and SQ counters per launch -- for the device loop a launch is a whole 30-iteration run) and the stamp breakdowns."""
import csv, glob, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF, OUT = os.path.join(ROOT, "gpurun_out", "prof_run"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
sha = open(os.path.join(PROF, "csrc_sha16.txt")).read().strip()
KERNEL = {"host": "map_score_kernel", "dev": "muse_loop_kernel"}


def counters(dirname, kernel):
    acc = {}
    for f in glob.glob(os.path.join(PROF, dirname, "*", "*_counter_collection.csv")):
        per = {}
        for row in csv.DictReader(open(f)):
            if kernel not in row["Kernel_Name"]:
                continue
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
        for (_, name), v in per.items():
            acc.setdefault(name, []).append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


rows = []
for L in ("host", "dev"):
    stats = glob.glob(os.path.join(PROF, "trace_" + L, "*", "*_kernel_stats.csv"))
    if len(stats) != 1:
        sys.exit(f"gpurun_out/prof_run/trace_{L}: expected one pass, found {len(stats)}")
    shutil.copy(stats[0], os.path.join(OUT, f"{tag}_muse_run_{L}_kernel_stats.csv"))
    k = [r for r in csv.DictReader(open(stats[0])) if KERNEL[L] in r["Name"]][0]
    c = {}
    for d in ("pmc_fetch_", "pmc_write_", "pmc_sq_"):
        c.update(counters(d + L, KERNEL[L]))
    fetch, write = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
    rows.append({"loop": L, "csrc_sha16": sha, "kernel": k["Name"][:120], "launches": k["Calls"], "avg_us_per_launch": float(k["AverageNs"]) / 1e3,
                 "min_us_per_launch": float(k["MinNs"]) / 1e3, "iterations_per_launch": 1 if L == "host" else 30,
                 "hbm_traffic_MB_per_launch": (2 * fetch + write) * 1024 / 1e6 if fetch is not None and write is not None else "",
                 **{n: c.get(n, "") for n in ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS",
                                               "GRBM_GUI_ACTIVE")}})
with open(os.path.join(OUT, f"{tag}_muse_run_summary.csv"), "w", newline="") as f:
    wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    wr.writeheader()
    wr.writerows(rows)
for name in ("runloop.log", "stamps_run_host.log", "stamps_run_dev.log"):
    src = os.path.join(PROF, name)
    if os.path.exists(src):
        with open(src) as f, open(os.path.join(OUT, f"{tag}_muse_run_{name}"), "w") as g:
            g.writelines(l for l in f if "amdgpu.ids" not in l)
for r in rows:
    print(r)
