#!/usr/bin/env python3
"""bench.py -- MC sims/sec (MAP+score) of the MUSE inner loop on MI355X.

One "step" = one pass of the hot path over one batch: for every one of the 512 simulations of the
batch, sample (x, z) ~ P(x, z | θ), find the latent MAP ẑ by L-BFGS/HagerZhang from a cold start
ẑ₀ = 0 to ||∇z||∞ <= 1e-2, and evaluate the score ∇θ logP(x, ẑ | θ) -- the get_J!/muse! map body of
the reference (src/muse.jl:169-176, :508-525), one launch per batch.  Workload at N=1 GPU:
BASELINE.json configs[1], Neal's funnel, 10^4-dim z, 1-dim θ (θ = 1), nsims = 512, fp64, synthetic
(Philox) data.  `python bench.py --gpus N` starts its own N ranks (one per GPU; under torch.distributed.run the ranks are
used as they come).  With N > 1 the map is sharded over the ranks and the per-rank score blocks are exchanged once per launch
-- both transports of the engine are measured in the same run (shared memory between the ranks of a node; RCCL all-gather)
and reported side by side with the ranks each communicator counts: --scaling strong (the default for N > 1; BASELINE.json's
north star) keeps the 512 sims of the step and gives every rank a contiguous block of 512/N of them; the steps are
independent maps, so a rank whose share is smaller than its GPU carries several consecutive steps in ONE launch
(maps_per_launch; MUSE_BENCH_MAPS=1: one step per launch, each element split over 2/4 workgroups instead, which is what a
dependent map -- a muse! iteration -- has to do); --scaling weak gives every rank its own 512-sim block of a 512*N-sim map.
Consecutive launches alternate over two lanes (streams with scratch and MAP slots of their own: MUSE_BENCH_LANES).

Prints ONE JSON line on rank 0 (see the driver contract), with two extra objects:
  roofline      the solver kernel against the bound that actually binds it, per launch: "hbm" -- compulsory HBM bytes of the
                placement the launch used / launch time against the 8 TB/s peak -- for the streaming placements; "valu" --
                ALGORITHMIC work (operation counts per element from the source x the measured issue cost of each kind) /
                (SIMDs x launch time x the clock measured inside the kernel) for the register/LDS resident placements, whose
                vectors never leave the chip, with the VALU-active-counter utilisation (rocprofv3, profiles/) beside it.
                Launch time = the pipelined step of the timed region (N = 1; `frac_kernel_time` is the same figure against the
                kernel's own mean duration between HIP events on the launch stream).  Both objects are
                always present (`roofline.hbm`, `roofline.valu`); SURVEY 8.d3's accounting figure is kept as
                `algorithmic_bytes_d3` and is not a roofline.
  cpu_baseline  the CPU oracle (oracle/, a restatement of the reference: "port") timed on this box's
                host cores on a bounded sample of the same workload (rank 0, N=1 only)
and, because the driver's record keeps only FLAT scalars of `config` / `roofline` / `cpu_baseline`:
  config.*      N = 1: the other BASELINE workloads (`funnel4_ms`, `noise_1e6_ms`, `smooth_1e5_ms` and their `_frac`: the pipelined
                step, two lanes -- one for the stencil model), the muse! iteration (`muse_iter_us`, `muse_iter_steady_us`,
                `muse_iter_line_search_us`), the 8-GPU share and the projections (`share_iter_us`, `proj_muse`, `proj_cfg4`, `proj_cfg5`),
                the score boards' hand-shake (`handshake_device`, `handshake_host`) -- flat_single();
                N > 1 (round 6): what was built for several GPUs, measured over THESE ranks beside the independent maps --
                `sharded_muse_iter_us` (muse_run_sharded, 30 iterations), `sharded_loop_ran` ("device" | "host" | "none"),
                `sharded_bit_equal` (every rank's trajectory against rank 0's unsharded muse_run), `handshake_*`,
                `cfg4_fd_H_sharded_ms`, `cfg5_smooth_1e5_sharded_ms` -- sharded_extras(), flat_sharded(); the full objects are under
                `extra` (N = 1) and `sharded` (N > 1)
  roofline.*    `frac` (ONE definition: work / (peak x the pipelined step)), `frac_kernel_time`, `hbm_frac`, `valu_util_frac`
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (model, N, ntheta, theta, nsims)
    "funnel_1e4": ("funnel", 10000, 1, [1.0], 512),        # BASELINE.json configs[1] (headline)
    "funnel_512": ("funnel", 512, 1, [1.0], 32),           # configs[0]
    "noise_1e6": ("noise", 1000000, 1, [0.5], 128),        # configs[2]
    "funnel4_1e4": ("funnel", 10000, 4, [1.0] * 4, 512),   # configs[3] (s/J pass)
    "smooth_1e5": ("smooth", 100000, 8, [1.0] * 8, 128),   # configs[4], per-GPU share
    "cfg5_smooth_1e5": ("smooth", 100000, 8, [1.0] * 8, 1024),   # configs[4], the whole job (shards under --gpus N)
    "cfg4_fd_H": ("funnel", 10000, 4, [1.0] * 4, 512),     # configs[3], get_H! by finite differences (shards under --gpus N)
}


HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 78.6    # fp64 vector peak: 256 CUs x 4 SIMDs x 16 FMA lanes/clk x 2 flop x 2.4 GHz
N_SIMD, CLOCK_HZ = 1024, 2.4e9
PROFILE_TAG = "r06"


def algorithmic_bytes(info, N):
    """SURVEY.md §8(d3) / BASELINE.md §3: words = 1 + 5E + Σ_k(4 h_k + 4) + 2 per sim, 8 B words.  An accounting
    figure that is independent of how much the kernel keeps on chip -- reported as `algorithmic_bytes_d3`."""
    E = info["f_calls"].astype(np.int64)
    K = info["iterations"].astype(np.int64)
    H = info["hist_words"].astype(np.int64)
    words = 1 + 5 * E + 4 * H + 4 * K + 2
    return int(8 * N * words.sum())


def compulsory_bytes(info, N, placement):
    """HBM bytes a launch MUST move in the placement it ran in (cold start from zero(z), converged solves), from the
    kernel's own per-sim counters E = f_calls, K = iterations, H = Σ_k h_k (history pairs used):
      resident  (z, s in registers, x, g in registers/LDS): zhat out; per kept iteration the pair (dx, dg) out;
                the two-loop recursion reads 4 history vectors per pair used:        1 + 2 (K-1) + 4 H
      streaming, elementwise models: the sampler pass also makes the initial evaluation and the first trial and
                writes x, s (2); every further trial reads z, s, x (3; from the unwritten zero start it reads s, x and
                writes z + c s into the MAP slot -- 3 -- which is the solve's output when that trial is the accepted last step:
                no last update pass then); a kept update reads z, s, x, g and writes z, dx, dg, g, s (9; the first one has neither z
                nor g to read: 7); the last update reads z, s, x and writes z (4; 3 from the zero start);
                8 words per history pair used (q is read and written once per pair in this placement)
      streaming, stencil model: sampler writes g (the true z, staged), x, z (3) and the A z pass reads g, x and
                writes x (3); initial evaluation reads z, x, writes g, s (4); a trial reads z, s, x (3); a kept
                update is two passes, 4 + 7 = 11; the last update 4; 8 per history pair used.
                "stencil_lds" (clusters whose members keep the search direction in LDS): every read or write of s
                drops out -- 3, 2, 3 + 6 = 9, 3 -- and the two-loop recursion moves 4 words per pair."""
    E = info["f_calls"].astype(np.int64)
    K = info["iterations"].astype(np.int64)
    H = info["hist_words"].astype(np.int64)
    kept = np.maximum(K - 1, 0)
    if placement == "resident":
        words = 1 + 2 * kept + 4 * H
    elif placement == "stencil":
        words = 6 + 4 + 3 * np.maximum(E - 1, 0) + 11 * kept + 4 * (K > 0) + 8 * H
    elif placement == "stencil_lds":  # the search direction (and the two-loop recursion's q) in LDS: one word less wherever
        # s was read or written, 4 instead of 8 per history pair
        words = 6 + 3 + 2 * np.maximum(E - 1, 0) + 9 * kept + 3 * (K > 0) + 4 * H
    else:
        first = (K == 1)  # the solve ended with its first line search: z stayed virtual until the last update
        trials = np.maximum(E - 2, 0)
        # (round 5: a further trial from the virtual zero also writes z + c s into the MAP slot -- 3 words instead of 2 -- and when
        #  it is the accepted step that ends the solve, which it is for every converged one-iteration solve with E >= 3, the last
        #  update pass is not run at all)
        words = np.where(first, np.where(trials >= 1, 2 + 3 * trials, 2 + 3),
                         2 + 2 * np.minimum(trials, 1) + 3 * np.maximum(trials - 1, 0) + 7 + 9 * np.maximum(kept - 1, 0)
                         + 4 * (K > 0) + 8 * H)
    return int(8 * N * words.sum())


# ---- algorithmic VALU work of the resident placements (the vectors never leave the chip: the bound is instruction issue) ----
# Operation counts per ELEMENT, read off the source (csrc/rng.hpp, models.hpp, solver.hpp), not off the compiler's output:
#   sampler (one Philox4x32-10 call -> two uniforms -> Box-Muller pair -> the model's (z, x)):
#     Philox: 10 rounds x 2 32x32->64 multiplies, 10 x 4 xors;  bits -> uniforms: 4 alignbit, 5 fp64;
#     log_unit: 29 fp64 (8 of them the IEEE division f/(2+f)) + 8 integer ops on the exponent/mantissa words;
#     sqrt_normal 10 + 1 (the factor -2);  sincospi_02: 29 fp64 + 8 integer/select;  Box-Muller products 2;  Model::sample 2
#   evaluation passes (funnel/noise gradient: 5 fp64; + g.s fma, |g| max):
#     sampler-fused initial evaluation AND first trial 14;  every further line-search trial 8;
#     last update pass (z += alpha s, step norm, score term) 5;  a kept update pass (K > 1) 14
# Issue cost per wave-instruction on one SIMD, measured by tools/clockprobe.hip with all CUs busy (HISTORY.md §6): fp64 4.4
# cycles (nominal 4: 16 lanes/clk), v_mad_u64_u32 4.75, 32-bit integer / select 2.7.
ALG_OPS = {"sampler_fp64": 5 + 29 + 11 + 29 + 2 + 2, "sampler_mul64": 20, "sampler_int32": 40 + 4 + 8 + 8,
           "init_fp64": 14, "trial_fp64": 8, "last_fp64": 5, "kept_fp64": 14}
ISSUE_CYCLES = {"fp64": 4.4, "mul64": 4.75, "int32": 2.7}


def algorithmic_valu(info, N, clock_hz):
    """(issue cycles per launch summed over the SIMDs, fp64 ops per launch) that the solves of `info` need by the counts
    above: cold start from zero(z), E = f_calls evaluations, K = iterations."""
    E = info["f_calls"].astype(np.float64)
    K = info["iterations"].astype(np.float64)
    fp64 = ALG_OPS["sampler_fp64"] + ALG_OPS["init_fp64"] + ALG_OPS["trial_fp64"] * np.maximum(E - 2, 0) \
        + ALG_OPS["last_fp64"] * (K > 0) + ALG_OPS["kept_fp64"] * np.maximum(K - 1, 0)
    cyc = fp64 * ISSUE_CYCLES["fp64"] + ALG_OPS["sampler_mul64"] * ISSUE_CYCLES["mul64"] + ALG_OPS["sampler_int32"] * ISSUE_CYCLES["int32"]
    elems = float(N)
    return float(cyc.sum() * elems / 64.0), float(fp64.sum() * elems)


def roofline_object(workload, model, N, info, pinfo, kernel_ms, launch_s, clock_hz_measured, lanes, ms_per_launch_pipelined, prow_ok=True):
    """The `roofline` object of a launch of `workload`: the solver kernel against the bound that binds the placement it ran in
    (module docstring).  launch_s: the time a launch is charged with -- the pipelined step of the timed region at N = 1 (launches
    complete once per step; with two lanes a launch's own duration between HIP events includes waiting for the compute units the
    launch before it still holds, and the event pair itself costs ~3 us), the kernel's mean duration otherwise."""
    mean_kernel_s = float(kernel_ms.mean()) * 1e-3
    placement = "resident" if pinfo["resident"] else (("stencil_lds" if pinfo["direction_in_lds"] else "stencil")
                                                      if model == "smooth" else "streaming")
    comp_bytes = compulsory_bytes(info, N, placement)
    prow, why = profile_row(workload) if prow_ok else (None, "profiles/ hold the 1-GPU, unsplit, unsharded launch")
    traffic = measured_traffic(workload) if prow is not None else None
    hbm = {"bound": "hbm", "achieved": comp_bytes / launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": comp_bytes / launch_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
           "compulsory_bytes_per_launch": comp_bytes, "launch_s_used": launch_s,
           "traffic_GBps": None if traffic is None else traffic / launch_s / 1e9}
    clock_hz = clock_hz_measured or CLOCK_HZ
    valu = None
    if placement == "resident":
        alg_cycles, alg_fp64 = algorithmic_valu(info, N, clock_hz)
        alg = {"fp64_ops_per_launch": alg_fp64, "issue_cycles_per_launch": alg_cycles,
               "frac_of_issue_peak": alg_cycles / (N_SIMD * launch_s * clock_hz),
               "fp64_TFLOPs_fma_equiv": 2.0 * alg_fp64 / launch_s / 1e12,
               "ops_per_element": ALG_OPS, "issue_cycles_per_wave_instruction": ISSUE_CYCLES,
               "note": "work / peak: operation counts per element from the source (bench.py: ALG_OPS) x the measured issue cost "
                       "of each kind (tools/clockprobe.hip) / (1024 SIMDs x launch time x measured clock); independent of how "
                       "many instructions the compiled kernel spends on them.  The peak is the PROBE's rate -- ~10 % softer than "
                       "the spec rates (fp64 4 cycles, 32-bit 2) and optimistic for this mix: the generator as a kernel of its own "
                       "at eight waves per SIMD (built, measured and removed in round 4: HISTORY.md section 6) ran no faster than "
                       "inside this kernel, i.e. the sampler -- two thirds of the launch -- already runs at the VALU's throughput"}
        valu = {"bound": "valu", "achieved": alg["frac_of_issue_peak"] * VALU_PEAK_TFLOPS, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": alg["frac_of_issue_peak"], "traffic": traffic, "algorithmic": alg,
                "clock_hz": clock_hz, "clock_source": "in-kernel s_memtime / s_memrealtime" if clock_hz_measured else "assumed",
                "launch_s_used": launch_s}
        if prow is not None and prow.get("SQ_ACTIVE_INST_VALU"):
            # VALU-active time of one launch: SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip's SIMDs
            # (MI355X_MICROARCH.md, cycle constants: SQ_ACTIVE_INST_* are in quad-cycles); the instruction stream of a
            # launch is fixed by its inputs, so the counter of the profiled launch is this launch's.
            busy = 4.0 * float(prow["SQ_ACTIVE_INST_VALU"])
            valu["utilisation"] = {"valu_active_cycles_per_launch": busy, "valu_insts_per_launch": float(prow["SQ_INSTS_VALU"]),
                                   "frac": busy / (N_SIMD * launch_s * clock_hz),
                                   "note": "VALU-active cycles (rocprofv3 SQ_ACTIVE_INST_VALU of profiles/) / (1024 SIMDs x launch "
                                           "time x measured clock): how busy the kernel's OWN instruction stream keeps the SIMDs "
                                           "(round 2 quoted this figure, against an assumed 2.4 GHz, as the roofline fraction)"}
    primary = dict(valu if (placement == "resident" and valu is not None) else hbm)
    primary.update({
        "kernel": "map_score_kernel", "placement": placement, "placement_info": pinfo, "kernel_ms_mean": 1e3 * mean_kernel_s,
        "lanes": lanes, "ms_per_launch_pipelined": ms_per_launch_pipelined,
        "kernel_ms_min": float(kernel_ms.min()), "launches_timed": int(kernel_ms.size),
        "algorithmic_bytes_d3": algorithmic_bytes(info, N),
        # ONE definition of `frac` across rounds: the launch is charged with the PIPELINED step of the timed region (launches complete
        # once per step; what rounds 3-5 reported in effect).  The same figure against the kernel's own mean duration between HIP events
        # (with two lanes that includes waiting for the compute units the launch before still holds) is beside it, never mixed in.
        "frac_definition": "pipelined step of the timed region" if abs(launch_s - mean_kernel_s) > 1e-12 else "mean kernel duration between HIP events",
        "frac_kernel_time": (valu if (placement == "resident" and valu is not None) else hbm)["frac"] * launch_s / mean_kernel_s,
        "hbm_frac": hbm["frac"],
        "valu_util_frac": (valu or {}).get("utilisation", {}).get("frac"),
        "hbm": hbm, "valu": valu,
        "profile": f"profiles/{PROFILE_TAG}_summary.csv" if prow is not None else None,
        "profile_note": why,
        "note": ("resident placement: z, s, x, g never leave registers/LDS, the HBM leg only carries zhat out (and the "
                 "L-BFGS pairs of solves with K > 1); the binding resource is fp64 VALU issue (sampler + evaluation passes). "
                 "frac = algorithmic work / issue peak (valu.algorithmic); valu.utilisation is how busy the compiled kernel keeps "
                 "the SIMDs.  BASELINE.json's '>= 40 % of the HBM roofline' can be neither met nor missed by this placement: it "
                 "moves 1 word per element where SURVEY 8.d3's accounting assumes 22 (hbm.frac is that one word / 8 TB/s)"
                 if placement == "resident" else
                 "streaming placement: achieved = compulsory bytes of the passes the solves made / kernel time"),
        "per_sim": {"f_calls_mean": float(info["f_calls"].mean()), "iterations_mean": float(info["iterations"].mean()),
                    "hist_pairs_mean": float(info["hist_words"].mean())},
    })
    return primary


def csrc_fingerprint():
    """sha256 over the kernel sources: profiles/ numbers are quoted only for the sources they were measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "museinference.jl_amd", "csrc")
    host_only = ("switches.hpp", "shm_gather.hpp")   # (headers of muse_engine.cpp / muse_comm.cpp alone: no kernel includes them)
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")) and f not in host_only:  # device code only: muse_kernels.hip and the headers it includes
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def profile_row(workload):
    """The workload's row of profiles/<tag>_summary.csv (rocprofv3 passes of this same command: tools/profile.sh,
    condensed by tools/summarize_profiles.py), or (None, reason).  A row measured on other kernel sources, or on
    another kernel than the launch used, is NOT quoted."""
    import csv
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_summary.csv")
    if not os.path.exists(path):
        return None, f"profiles/{PROFILE_TAG}_summary.csv not found"
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["workload"] == workload:
                if row.get("csrc_sha16") != csrc_fingerprint():
                    return None, (f"profiles/{PROFILE_TAG}_summary.csv was measured on kernel sources {row.get('csrc_sha16')}, "
                                  f"this tree is {csrc_fingerprint()}: re-run tools/profile.sh")
                return row, None
    return None, "workload not profiled"


def measured_traffic(workload):
    """HBM bytes per solver launch from the rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs of this
    same command; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB, the factor 2 being the gfx950 FETCH_SIZE correction of
    MI355X_MICROARCH.md §HBM).  None (with the reason on stderr) if there is no current profile."""
    row, why = profile_row(workload)
    if row is None:
        if why != "workload not profiled":
            print(f"[bench] roofline.traffic unavailable: {why}", file=sys.stderr)
        return None
    try:
        return float(row["hbm_traffic_MB"]) * 1e6
    except (KeyError, ValueError):
        return None


def usable_cores(limit):
    """Threads the CPU baseline may really use: affinity mask and the cgroup CPU quota (cpu.max), whichever is smaller."""
    n = limit
    if hasattr(os, "sched_getaffinity"):
        n = min(n, len(os.sched_getaffinity(0)))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(model, N, theta, seed, cpu_seconds=30.0):
    """Time the oracle on the host cores over a bounded number of sims of the same workload
    (about `cpu_seconds` of CPU work in total, OpenMP over sims on every core the process may use)."""
    from oracle import oracle as O
    O.build()
    cores = usable_cores(O.num_threads())
    n1 = 32
    O.map_and_score_batch(model, N, seed, 0, 4, theta, atol=1e-2, z0_mode=0, nthreads=1)       # warm the library
    t0 = time.perf_counter()
    O.map_and_score_batch(model, N, seed, 0, n1, theta, atol=1e-2, z0_mode=0, nthreads=1)
    t1 = (time.perf_counter() - t0) / n1  # seconds per sim, one thread
    O.map_and_score_batch(model, N, seed, 0, 4 * cores, theta, atol=1e-2, z0_mode=0, nthreads=cores)  # spin the team up
    nall = int(max(8 * cores, cpu_seconds / max(t1, 1e-9)))
    t0 = time.perf_counter()
    O.map_and_score_batch(model, N, seed, 0, nall, theta, atol=1e-2, z0_mode=0, nthreads=cores)
    tall = time.perf_counter() - t0
    out = {
        "value": nall / tall, "unit": "sims/s", "cores": cores, "kind": "port",
        "sample": f"{nall} sims of the same workload (oracle/muse_oracle.c, gcc -O3, OpenMP over sims, "
                  f"{cores} threads, {tall:.2f} s wall); 1 thread: {1.0 / t1:.1f} sims/s on {n1} sims",
        "value_1thread": 1.0 / t1,
    }
    # A second figure, against code a CPU user would actually run (-O3 -march=native): the same source built on
    # THIS machine with contraction and re-association allowed and the generator written for SIMD lanes (oracle/Makefile,
    # target `fast`).  Not a checker: its scores are compared with the strict build's at a tolerance, never the other way.
    try:
        g_ref, _, i_ref = O.map_and_score_batch(model, N, seed, 0, 16, theta, atol=1e-2, z0_mode=0, nthreads=1)
        with O.fast_build():
            g_f, _, i_f = O.map_and_score_batch(model, N, seed, 0, 16, theta, atol=1e-2, z0_mode=0, nthreads=1)
            t0 = time.perf_counter()
            O.map_and_score_batch(model, N, seed, 0, n1, theta, atol=1e-2, z0_mode=0, nthreads=1)
            t1f = (time.perf_counter() - t0) / n1
            O.map_and_score_batch(model, N, seed, 0, 4 * cores, theta, atol=1e-2, z0_mode=0, nthreads=cores)
            nallf = int(max(8 * cores, 0.5 * cpu_seconds / max(t1f, 1e-9)))
            t0 = time.perf_counter()
            O.map_and_score_batch(model, N, seed, 0, nallf, theta, atol=1e-2, z0_mode=0, nthreads=cores)
            tallf = time.perf_counter() - t0
        rel = float(np.max(np.abs(g_f - g_ref) / np.maximum(np.abs(g_ref), 1e-300)))
        out["vectorised"] = {
            "value": nallf / tallf, "value_1thread": 1.0 / t1f, "unit": "sims/s", "cores": cores,
            "flags": "-O3 -march=native -ffp-contract=fast -fassociative-math -fno-signed-zeros -fno-trapping-math -fno-math-errno",
            "max_rel_score_difference_vs_strict_build": rel, "same_evaluation_counts": bool(np.array_equal(i_f["f_calls"], i_ref["f_calls"])),
            "sample": f"{nallf} sims, {cores} threads, {tallf:.2f} s wall",
        }
        if not (rel < 1e-9):
            out["vectorised"]["note"] = "scores differ from the strict build beyond 1e-9: figure not comparable"
    except Exception as e:  # noqa: BLE001 -- a second figure, never at the cost of the line
        out["vectorised"] = {"skipped": f"{type(e).__name__}: {e}"}
    return out


def pipelined_steps(M, prob, seed, sim_lo, sim_hi, theta, K, areas=4, outs=None):
    """K independent cold-start maps over [sim_lo, sim_hi), software-pipelined over the result areas (and the lanes the context
    was given); returns the last step's (scores, infos)."""
    n = sim_hi - sim_lo
    if outs is None:
        outs = [(np.empty((n, prob.ntheta)), np.zeros(n, dtype=M._capi.INFO_DTYPE)) for _ in range(areas)]
    pend, last = [], None
    for k in range(K):
        prob.map_and_score_batch_async(seed, sim_lo, sim_hi, theta, atol=1e-2, z0_mode=M.Z0_ZERO, result_area=k % areas)
        pend.append(k % areas)
        if len(pend) > areas - 1:
            a = pend.pop(0)
            last = prob.batch_wait(n, a, out=outs[a])
    while pend:
        a = pend.pop(0)
        last = prob.batch_wait(n, a, out=outs[a])
    return last


def quick_workload(M, name, device, seconds=0.3, seed=0):
    """One of the other BASELINE workloads, timed briefly inside the default run so that it is in the driver's line: the same
    pipelined cold-start steps as the headline's timed region (normals cache off, two lanes except the stencil model), for about
    `seconds`, then a short leg with HIP events around every launch for the roofline object (bound, frac, traffic from profiles/)."""
    model, N, nth, theta, nsims = WORKLOADS[name]
    prob = M.HipMuseProblem(None, model=model, ntheta=nth, N=N, device=device)
    try:
        prob.set_normals_cache(False)
        lanes = 1 if model == "smooth" else 2
        if lanes > 1:
            prob.set_concurrency(lanes)
        outs = [(np.empty((nsims, nth)), np.zeros(nsims, dtype=M._capi.INFO_DTYPE)) for _ in range(4)]
        pipelined_steps(M, prob, seed, 0, nsims, theta, 3, outs=outs)
        prob.synchronize()
        t0 = time.perf_counter()
        pipelined_steps(M, prob, seed, 0, nsims, theta, 4, outs=outs)
        prob.synchronize()
        t_step = (time.perf_counter() - t0) / 4
        K = int(max(8, min(4000, seconds / max(t_step, 1e-6))))
        t0 = time.perf_counter()
        g, info = pipelined_steps(M, prob, seed, 0, nsims, theta, K, outs=outs)
        prob.synchronize()
        dt = (time.perf_counter() - t0) / K
        nprof = 8 if t_step > 5e-4 else 32
        prob.profile_begin(nprof + 8)
        pipelined_steps(M, prob, seed, 0, nsims, theta, nprof, outs=outs)
        kernel_ms = prob.profile_end()
        try:
            clock_hz = prob.profile_clock_hz()
        except M.MuseError:
            clock_hz = None
        assert np.all(info["status"] == 0), "a MAP solve did not converge"
        pinfo = prob.placement_info()
        launch_s = dt   # the pipelined step (the headline's definition)
        roof = roofline_object(name, model, N, info, pinfo, kernel_ms, launch_s, clock_hz, lanes, 1e3 * dt)
        return {"ms_per_step": 1e3 * dt, "sims_per_s": nsims / dt, "steps_timed": K, "nsims": nsims, "N": N, "ntheta": nth,
                "bound": roof["bound"], "frac": roof["frac"], "frac_kernel_time": roof["frac_kernel_time"], "traffic": roof["traffic"], "achieved": roof["achieved"],
                "unit": roof["unit"], "kernel_ms_mean": roof["kernel_ms_mean"], "lanes": lanes, "placement": roof["placement"],
                "compulsory_bytes_per_launch": roof["hbm"]["compulsory_bytes_per_launch"], "profile": roof["profile"],
                "per_sim": roof["per_sim"]}
    finally:
        prob.close()


def iteration_regimes(hist, info):
    """The per-iteration times a native loop records (hist[:, -1], seconds), grouped by what its solves did: "line_search" --
    (nearly) every element's solve took a line search (f_calls = 3: what every iteration of a run with the reference's default
    theta_rtol looks like) -- and "converged_at_start" -- theta has stopped moving, the warm starts pass the gradient test at once
    (f_calls = 1), which only a run driven far past convergence (theta_rtol = 1e-12, as these timing runs are) ever reaches.  The
    first iteration (cold start: the generator) belongs to neither."""
    out = {}
    fc = info["f_calls"].mean(axis=1)
    t = 1e6 * hist[:, -1]
    for name, sel in (("line_search", fc >= 2.5), ("converged_at_start", fc <= 1.05)):
        sel = sel.copy()
        sel[0] = False
        if sel.any():
            out[name] = {"us_per_outer_iteration": float(np.median(t[sel])), "iterations": int(sel.sum())}
    return out


def share_rates(M, xdata, model, nth, nsims, seed, device, whole_job_steady_us, ngpus=8, whole_job_regimes=None):
    """What ONE rank of an 8-GPU job runs per outer iteration of muse! (a DEPENDENT map sequence, src/muse.jl:159-232): the
    native sharded loop (muse_run_sharded) at nsims/8 + 1 elements, measured on this one GPU with a one-rank shared-memory
    communicator -- the gathered map, the hand-off through the segment, the step on the host.  The ratio to the whole job's
    iteration on one GPU is the speed-up 8 GPUs can give that loop at best (the exchange with 8 ranks costs at least what it
    costs with one)."""
    share = max(2, nsims // ngpus)
    out = {"elements_per_rank": share + 1, "ngpus": ngpus}
    try:
        prob = M.HipMuseProblem(xdata, model=model, ntheta=nth, device=device, prior=M.GaussianPrior(0.0, 3.0))
        kw = dict(nsims=share, maxsteps=30, theta_rtol=1e-12, atol=1e-2, alpha=0.7)
        for name, dev in (("device_loop_no_exchange", True), ("host_loop_no_exchange", False)):
            best = float("inf")
            for _ in range(3):
                t0 = time.perf_counter()
                n, _, hist, _, info = prob.run_muse(seed, [1.0] * nth, device_loop=dev, **kw)
                best = min(best, (time.perf_counter() - t0) / max(1, n))
            out[name] = {"us_per_outer_iteration_30": 1e6 * best, "us_per_outer_iteration_steady": 1e6 * float(np.median(hist[5:, -1])),
                         "by_regime": iteration_regimes(hist, info)}
        prob.comm_init(1, 0, M.HipMuseProblem.comm_unique_id("shm", 4096))
        P = M.HipMuseProblem
        for name, flags in (("sharded_loop_shm_1rank", 0), ("sharded_loop_host_board_shm_1rank", P.DEBUG_HOST_BOARD),
                            ("sharded_host_loop_shm_1rank", P.DEBUG_SHARDED_HOST_LOOP)):
            # the sharded loop as the library runs it -- ONE persistent launch per rank, the ranks' scores meeting on a board per GPU in
            # device memory that every rank maps (hipIpc; round 5) --, the same with the ONE board in pinned host memory (what runs where
            # the ranks cannot map each other's device memory), and the host-driven loop of round 4 (gathered map, step on the host)
            prob.debug_flags(flags)
            best = float("inf")
            for _ in range(3):
                t0 = time.perf_counter()
                n, _, hist, _, info = prob.run_muse_sharded(seed, [1.0] * nth, **kw)
                best = min(best, (time.perf_counter() - t0) / max(1, n))
            out[name] = {"us_per_outer_iteration_30": 1e6 * best, "us_per_outer_iteration_steady": 1e6 * float(np.median(hist[5:, -1])),
                         "by_regime": iteration_regimes(hist, info), "loop_ran": prob.comm_board_status()["last_loop"]}
        prob.debug_flags(0)
        out["board_handshake"] = prob.comm_board_status()
        prob.close()
        if whole_job_steady_us:
            out["projected_speedup_at_8_gpus"] = whole_job_steady_us / out["sharded_loop_shm_1rank"]["us_per_outer_iteration_steady"]
        if whole_job_regimes:
            proj = {}
            for reg, w in whole_job_regimes.items():
                sh = out["sharded_loop_shm_1rank"]["by_regime"].get(reg)
                if sh:
                    proj[reg] = {"whole_job_us": w["us_per_outer_iteration"], "share_us": sh["us_per_outer_iteration"],
                                 "projected_speedup_at_8_gpus": w["us_per_outer_iteration"] / sh["us_per_outer_iteration"]}
            out["projected_by_regime"] = proj
        out["note"] = ("a muse! iteration is ONE dependent map: per rank nsims/8 + 1 one-workgroup problems (one round on 256 CUs: one "
                       "problem's latency), then the exchange and the step before the next map can start.  sharded_loop_shm_1rank: "
                       "muse_run_sharded as the library runs it -- a persistent loop kernel per rank, scores stored into every rank's board "
                       "in device memory (hipIpc mappings), no host between two maps -- measured with ONE rank in the communicator (an "
                       "upper bound for 8: stores into other GPUs' boards cross xGMI; ..._host_board_...: the one board in pinned host memory, "
                       "a PCIe round trip per iteration whoever writes it).  projected_speedup_at_8_gpus = the whole "
                       "job's steady iteration on one GPU / this rank's steady iteration, both the median from iteration 6 on of a "
                       "30-iteration run -- which mixes two regimes (iteration_regimes): projected_by_regime compares like with like; "
                       "line_search is what every iteration of a run with the reference's default theta_rtol is")
    except Exception as e:  # noqa: BLE001 -- an extra: never at the cost of the line
        out["skipped"] = f"{type(e).__name__}: {e}"
    return out


def extra_rates(M, sampler, model, N, nth, theta, nsims, seed, device):
    """SURVEY.md §8(d1) asks for three rates; the headline (value) is the get_J!-style cold-start pass.  The other
    two, measured here OUTSIDE the timed region on a problem with observed data: the steady-state muse! map
    (nsims+1 elements, warm starts from the previous iteration's MAPs, src/muse.jl:169-181) and the get_H!
    finite-difference map (1 fiducial + 2 nθ perturbed MAP+score per sim, src/muse.jl:407-446); plus the wall
    time of a complete muse() run (host algebra included)."""
    # observed data: a draw at theta_true = 0 (sampled with the bench's own problem object: creating and destroying one
    # more context here was seen to put a ~100-launch stretch of the loops below at 8x the step time inside the runtime)
    xdata, _ = sampler.sample_x_z(M.SimRng(seed, M.DATA_SIM), [0.0] * nth)
    prob = M.HipMuseProblem(xdata, model=model, ntheta=nth, device=device, prior=M.GaussianPrior(0.0, 3.0))
    out = {}
    prob.map_and_score_batch(seed, 0, nsims, theta, include_data=True, z0_mode=M.Z0_ZERO)  # iteration 1: cold
    K, areas = 100, 4
    prob.set_timing(False)
    best = float("inf")
    for _ in range(3):  # fastest of three loops
        prob.synchronize()
        t0 = time.perf_counter()
        pend = []
        for k in range(K):
            n = prob.map_and_score_batch_async(seed, 0, nsims, theta, include_data=True, z0_mode=M.Z0_WARM,
                                               result_area=k % areas)
            pend.append((n, k % areas))
            if len(pend) > areas - 1:
                prob.batch_wait(*pend.pop(0))
        while pend:
            prob.batch_wait(*pend.pop(0))
        prob.synchronize()
        best = min(best, time.perf_counter() - t0)
    out["muse_map_warm_sims_per_s"] = (nsims + 1) * K / best
    out["muse_map_warm_us_per_step"] = 1e6 * best / K
    prob.set_timing(True)
    nH = max(1, nsims // 8)
    step = [0.05] * nth
    prob.fd_jacobian_batch(seed, 0, nH, theta, step)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        prob.fd_jacobian_batch(seed, 0, nH, theta, step)
    dt = time.perf_counter() - t0
    out["get_H_fd_maps_per_s"] = reps * (1 + 2 * nth * nH) / dt
    out["get_H_fd_nsims"] = nH
    M.muse(prob, [1.0] * nth, rng=seed, nsims=nsims, get_covariance=True)  # first call: pinned result areas are allocated
    t0 = time.perf_counter()
    res = M.muse(prob, [1.0] * nth, rng=seed, nsims=nsims, get_covariance=True)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    res30 = M.muse(prob, [1.0] * nth, rng=seed, nsims=nsims, maxsteps=30, theta_rtol=1e-12)
    dt30 = time.perf_counter() - t0
    # the native loops alone (no Python between the iterations, no history records built): ONE launch for all iterations
    # (the default) / one launch per iteration with the algebra on the host
    loops, steady, regimes = {}, {}, {}
    for name, dev in (("host_loop", False), ("device_loop", True)):
        best = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            n30, _, hist30, _, info30 = prob.run_muse(seed, [1.0] * nth, nsims=nsims, maxsteps=30, theta_rtol=1e-12, atol=1e-2, alpha=0.7, device_loop=dev)
            best = min(best, (time.perf_counter() - t0) / max(1, n30))
        loops[name] = 1e6 * best
        steady[name] = 1e6 * float(np.median(hist30[5:, -1])) if n30 > 6 else None
        regimes[name] = iteration_regimes(hist30, info30)
    out["muse_run"] = {"wall_s": dt, "outer_iterations": len(res.history), "theta": [float(t) for t in res.theta],
                       "sigma": [float(t) for t in np.sqrt(np.diag(np.atleast_2d(res.Sigma)))],
                       "us_per_outer_iteration_30": loops["device_loop"],
                       "us_per_outer_iteration_30_host_loop": loops["host_loop"],
                       "us_per_outer_iteration_steady": steady["device_loop"],
                       "us_per_outer_iteration_steady_host_loop": steady["host_loop"],
                       "by_regime": regimes["device_loop"], "by_regime_host_loop": regimes["host_loop"],
                       "us_per_outer_iteration_30_with_python_history": 1e6 * dt30 / max(1, len(res30.history)),
                       "note": "muse(prob, theta0=1; nsims, get_covariance=True): outer iterations (native loop) + get_J! + get_H!, "
                               "host algebra included; us_per_outer_iteration_30: wall of a 30-iteration muse_run_device call (what "
                               "muse() runs: ONE launch for all iterations, scores exchanged between the workgroups as tagged granules, "
                               "the step on the GPU) per iteration, the first (cold: generator) iteration and the call's own launch and "
                               "copy-out included; ..._host_loop: muse_run, one launch per iteration, algebra on the host; "
                               "..._steady: the median of the per-iteration times the loop itself records from iteration 6 on; "
                               "..._with_python_history: through muse(), which also builds the 30 history records"}
    out["muse_run_8gpu_share"] = share_rates(M, xdata, model, nth, nsims, seed, device, steady["device_loop"],
                                             whole_job_regimes=regimes["device_loop"])
    prob.close()
    # the get_H! finite-difference map at configs[3]'s OWN shape: funnel, N = 10^4, 4 theta blocks, 512 sims ->
    # 1 fiducial + 512 x 4 x 2 perturbed MAP+score problems (src/muse.jl:407-446); last, on a context of its own
    p4 = M.HipMuseProblem(None, model="funnel", ntheta=4, N=10000, device=device)
    th4, st4 = [1.0] * 4, [0.05] * 4
    p4.fd_jacobian_batch(seed, 0, 512, th4, st4)
    best = float("inf")
    for _ in range(5):
        t0 = time.perf_counter()
        p4.fd_jacobian_batch(seed, 0, 512, th4, st4)
        best = min(best, time.perf_counter() - t0)
    out["get_H_fd_configs3"] = {"maps_per_s": (1 + 2 * 4 * 512) / best, "ms_per_call": 1e3 * best, "nsims": 512, "ntheta": 4,
                                "maps_per_call": 1 + 2 * 4 * 512}
    p4.close()
    return out


# ---- the workloads BASELINE.json puts on 8 GPUs (configs[3], configs[4]) --------------------------------------------------------------
# cfg4_fd_H: get_H!'s finite-difference map (src/muse.jl:407-446) of the 4-theta funnel at N = 10^4 over 512 sims: 1 fiducial +
#            512 x 4 x 2 perturbed MAP+score problems per call, shared over the ranks as contiguous blocks of the flattened
#            (sim, column) list (the reference's rule of mapping over the longer axis, src/muse.jl:327-333, across ranks), one
#            exchange of the column blocks per call.
# cfg5_smooth_1e5: the s/J map (src/muse.jl:508-525) of the hierarchical linear-Gaussian field, N = 10^5, 8 theta, 1024 sims, shared
#            over the ranks as contiguous blocks of sims, one exchange of the score blocks per map.
FD_WORKLOAD = {"model": "funnel", "N": 10000, "ntheta": 4, "theta": [1.0] * 4, "step": [0.05] * 4, "nsims": 512}


def fd_call(M, prob, seed, lo, hi, sharded):
    """One get_H! finite-difference call over the (sim, column) units [lo, hi) of the 512 x 4 list, and -- sharded -- the exchange
    of the column block through the engine's communicator.  Returns (columns, infos)."""
    w = FD_WORKLOAD
    cols, info = prob.fd_jacobian_columns(seed, 0, lo, hi, w["theta"], w["step"])
    if sharded:
        prob.allgather_scores(cols.reshape(-1))
    return cols, info


def fd_problems(lo, hi, nth):
    """MAP+score problems of one call over units [lo, hi): two perturbed ones per unit and the shared fiducial one."""
    return 2 * (hi - lo) + 1


def scale_projection(M, device, seed=0, ngpus=8):
    """configs[3] and configs[4] -- the workloads BASELINE.json assigns to 8 GPUs -- on ONE GPU: the whole job, and one rank's
    share of it through a one-rank shared-memory communicator (the exchange included).  Both are maps whose elements keep every
    compute unit busy at the 8-GPU share too, so the ratio is the speed-up 8 GPUs can give (an upper bound: an 8-rank exchange
    costs at least what a 1-rank one does; it is one hand-off per map of tens to hundreds of microseconds to milliseconds)."""
    out = {}
    # ---- cfg4: FD get_H!
    try:
        w = FD_WORKLOAD
        nunits = w["nsims"] * w["ntheta"]
        prob = M.HipMuseProblem(None, model=w["model"], ntheta=w["ntheta"], N=w["N"], device=device)
        fd_call(M, prob, seed, 0, nunits, False)
        best = float("inf")
        for _ in range(5):
            t0 = time.perf_counter()
            fd_call(M, prob, seed, 0, nunits, False)
            best = min(best, time.perf_counter() - t0)
        whole = best
        lo, hi = M.block_partition(0, nunits, ngpus, 0)
        prob.comm_init(1, 0, M.HipMuseProblem.comm_unique_id("shm", 4096))
        fd_call(M, prob, seed, lo, hi, True)
        best = float("inf")
        for _ in range(8):
            t0 = time.perf_counter()
            fd_call(M, prob, seed, lo, hi, True)
            best = min(best, time.perf_counter() - t0)
        prob.close()
        out["cfg4_fd_H"] = {"whole_job_ms": 1e3 * whole, "share_ms": 1e3 * best, "projected_speedup_at_8_gpus": whole / best,
                            "problems_whole": fd_problems(0, nunits, w["ntheta"]), "problems_share": fd_problems(lo, hi, w["ntheta"]),
                            "maps_per_s_whole": fd_problems(0, nunits, w["ntheta"]) / whole,
                            "bound": "the share is 513 problems -- two rounds of the 256 compute units, as the whole job is eighteen -- plus the "
                                     "call's fixed cost: two launches, the upload of the sampling thetas, one exchange"}
    except Exception as e:  # noqa: BLE001 -- an extra: never at the cost of the line
        out["cfg4_fd_H"] = {"skipped": f"{type(e).__name__}: {e}"}
    # ---- cfg5: the 1024-sim map of the stencil model
    try:
        model, N, nth, theta, nsims = WORKLOADS["cfg5_smooth_1e5"]
        share = nsims // ngpus
        prob = M.HipMuseProblem(None, model=model, ntheta=nth, N=N, device=device)
        prob.set_normals_cache(False)
        outs = [(np.empty((nsims, nth)), np.zeros(nsims, dtype=M._capi.INFO_DTYPE)) for _ in range(4)]
        pipelined_steps(M, prob, seed, 0, nsims, theta, 1, outs=outs)
        prob.synchronize()
        t0 = time.perf_counter()
        pipelined_steps(M, prob, seed, 0, nsims, theta, 3, outs=outs)
        prob.synchronize()
        whole = (time.perf_counter() - t0) / 3
        prob.comm_init(1, 0, M.HipMuseProblem.comm_unique_id("shm", share * nth))
        outs = [(np.empty((1, share, nth)), np.zeros(share, dtype=M._capi.INFO_DTYPE)) for _ in range(4)]

        def gathered(K):
            pend = []
            for k in range(K):
                prob.map_and_score_batch_gather_async(seed, 0, share, theta, share, atol=1e-2, z0_mode=M.Z0_ZERO, result_area=k % 4)
                pend.append(k % 4)
                if len(pend) > 3:
                    a = pend.pop(0)
                    prob.batch_wait_gathered(share, share, a, out=outs[a])
            while pend:
                a = pend.pop(0)
                prob.batch_wait_gathered(share, share, a, out=outs[a])
        gathered(2)
        prob.synchronize()
        t0 = time.perf_counter()
        gathered(12)
        prob.synchronize()
        part = (time.perf_counter() - t0) / 12
        prob.close()
        out["cfg5_smooth_1e5"] = {"whole_job_ms": 1e3 * whole, "share_ms": 1e3 * part, "projected_speedup_at_8_gpus": whole / part,
                                  "sims_whole": nsims, "sims_share": share, "sims_per_s_whole": nsims / whole,
                                  "bound": "HBM on both sides: 1024 and 128 sims are 32 and 4 rounds of the 32 clusters of 16 workgroups a GPU holds (two workgroups per compute unit); the "
                                           "exchange is one 8 KB hand-off per 2 ms map"}
    except Exception as e:  # noqa: BLE001
        out["cfg5_smooth_1e5"] = {"skipped": f"{type(e).__name__}: {e}"}
    out["note"] = ("whole job on one GPU / one rank's share of it (1/8 of the units, through a one-rank shared-memory communicator, "
                   "exchange included): what 8 GPUs can give these maps at best; `python bench.py --gpus N --workload cfg4_fd_H | "
                   "cfg5_smooth_1e5` measures the same on N GPUs")
    return out


def sharded_extras(M, torch, dist, args, world, rank, local_rank, tdev, seed=0):
    """What the driver's N > 1 command (`bench.py --gpus N`, default arguments) times BESIDE the headline's independent maps, so that
    the first run on more than one GPU measures what was built for it -- every rank calls this (collective), about 3 s in all:
      muse_run   the DEPENDENT path: muse_run_sharded over the N ranks at configs[1] (30 iterations of src/muse.jl:159-232, the pmap of
                 :169 over the ranks, src/util.jl:74-83): us per iteration, WHICH loop ran (persistent launch through the boards in device
                 memory / the board in pinned host memory / host-driven), the verdict of the boards' set-up hand-shake, and a bit-compare of
                 every rank's theta trajectory and scores with rank 0's UNSHARDED muse_run of the same job;
      cfg4_fd_H  configs[3]: get_H! by finite differences (src/muse.jl:426-442), the (sim, column) list in N blocks + one exchange;
      cfg5_smooth_1e5  configs[4]: the 1024-sim s/J map of the stencil model (src/muse.jl:508-525) in N blocks + one exchange.
    --small (development aid; the multi-rank tests on ONE GPU): the same code at sizes eight processes can share a GPU with."""
    out = {}
    P = M.HipMuseProblem

    def uid_for(block_doubles):
        u = [P.comm_unique_id("shm", block_doubles) if rank == 0 else None]
        dist.broadcast_object_list(u, src=0)
        return u[0]

    def tmax(x):
        t = torch.tensor([x], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ok(flag):
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def sync(prob):
        prob.synchronize()
        dist.barrier()

    # ---- the dependent path: the sharded muse! loop at configs[1]
    try:
        model, N, nth, theta, nsims = WORKLOADS["funnel_1e4"]
        if args.small:
            nsims = 16 * world
        smp = P(None, model=model, ntheta=nth, N=N, device=local_rank)
        xdata, _ = smp.sample_x_z(M.SimRng(seed, M.DATA_SIM), [0.0] * nth)     # the same bits on every rank
        smp.close()
        prob = P(xdata, model=model, ntheta=nth, device=local_rank, prior=M.GaussianPrior(0.0, 3.0))
        prob.comm_init(world, rank, uid_for(max(4096, (nsims + 1) * nth)))
        seen = prob.comm_ranks_seen()
        t0 = time.perf_counter()
        hs = prob.comm_board_status()          # collective: maps the boards, proves each kind by the hand-shake
        t_setup = tmax(time.perf_counter() - t0)
        kw = dict(nsims=nsims, maxsteps=10 if args.small else 30, theta_rtol=1e-12, atol=1e-2, alpha=0.7)
        runs = {}
        for name, flags in (("default", 0), ("host_board", P.DEBUG_HOST_BOARD), ("host_loop", P.DEBUG_SHARDED_HOST_LOOP)):
            prob.debug_flags(flags)
            prob.run_muse_sharded(seed, [1.0] * nth, **kw)     # warm: buffers, the loop kernel's code object
            best, res = float("inf"), None
            for _ in range(3):
                sync(prob)
                t0 = time.perf_counter()
                res = prob.run_muse_sharded(seed, [1.0] * nth, **kw)
                best = min(best, tmax(time.perf_counter() - t0))
            n, th, hist, gs, info = res
            runs[name] = {"us_per_outer_iteration_30": 1e6 * best / max(1, n), "us_per_outer_iteration_steady": 1e6 * float(np.median(hist[5:, -1])) if n > 6 else None,
                          "iterations": int(n), "loop_ran": prob.comm_board_status()["last_loop"], "by_regime": iteration_regimes(hist, info)}
            if name == "default":
                mine = (n, th.copy(), hist[:, :-1].copy(), gs.copy())
        prob.debug_flags(0)
        # rank 0's unsharded loop of the same job (one launch per iteration, the step on the host: muse_run) -> every rank compares
        ref = [None]
        if rank == 0:
            one = P(xdata, model=model, ntheta=nth, device=local_rank, prior=M.GaussianPrior(0.0, 3.0))
            n1, t1, h1, g1, _ = one.run_muse(seed, [1.0] * nth, device_loop=False, **kw)
            one.close()
            ref[0] = (n1, t1, h1[:, :-1].copy(), g1)
        dist.broadcast_object_list(ref, src=0)
        n1, t1, h1, g1 = ref[0]
        same = mine[0] == n1 and np.array_equal(mine[1], t1) and np.array_equal(mine[2], h1) and np.array_equal(mine[3], g1)
        prob.close()
        out["muse_run"] = {"nsims": nsims, "N": N, "ranks_seen": seen, "board": hs["board"], "handshake": hs, "board_setup_ms": 1e3 * t_setup,
                           "runs": runs, "trajectory_bit_equal_to_unsharded_on_every_rank": all_ok(same), "theta": [float(t) for t in t1],
                           "note": "muse_run_sharded over the ranks of this job, 30 iterations driven past convergence (theta_rtol 1e-12); wall of "
                                   "the call / iterations, MAX over ranks, best of three; `default` is what the library runs by itself, the other two "
                                   "are the fall-backs forced by a debug flag; the compare is against rank 0's unsharded muse_run: theta per "
                                   "iteration, every record, every simulation's score, bit for bit"}
    except Exception as e:  # noqa: BLE001 -- an extra: never at the cost of the line (every rank fails or none: the calls are collective)
        out["muse_run"] = {"skipped": f"{type(e).__name__}: {e}"}
    # ---- configs[3]: get_H! by finite differences, sharded over the (sim, column) list
    try:
        w = dict(FD_WORKLOAD)
        if args.small:
            w["nsims"] = 8 * world
        nunits = w["nsims"] * w["ntheta"]
        prob = P(None, model=w["model"], ntheta=w["ntheta"], N=w["N"], device=local_rank)
        lo, hi = M.block_partition(0, nunits, world, rank)
        prob.comm_init(world, rank, uid_for(4096))

        def call():
            cols, info = prob.fd_jacobian_columns(seed, 0, lo, hi, w["theta"], w["step"])
            prob.allgather_scores(np.pad(cols.reshape(-1), (0, (-(-nunits // world)) * w["ntheta"] - cols.size)))
            return info
        call()
        K = 8
        sync(prob)
        t0 = time.perf_counter()
        for _ in range(K):
            info = call()
        sync(prob)
        dt = tmax(time.perf_counter() - t0) / K
        ok = all_ok(bool(np.all(info["status"] == 0)))
        prob.close()
        out["cfg4_fd_H"] = {"ms_per_call": 1e3 * dt, "problems_per_call": 2 * nunits + 1, "maps_per_s": (2 * nunits + 1) / dt, "converged_on_every_rank": ok,
                            "units_per_rank": hi - lo, "nsims": w["nsims"]}
    except Exception as e:  # noqa: BLE001
        out["cfg4_fd_H"] = {"skipped": f"{type(e).__name__}: {e}"}
    # ---- configs[4]: the stencil model's s/J map, sharded over the sims
    try:
        model, N, nth, theta, nsims = WORKLOADS["cfg5_smooth_1e5"]
        if args.small:
            N, nsims = 20000, 4 * world
        lo, hi = M.block_partition(0, nsims, world, rank)
        rows = -(-nsims // world)
        prob = P(None, model=model, ntheta=nth, N=N, device=local_rank)
        prob.set_normals_cache(False)
        prob.comm_init(world, rank, uid_for(rows * nth))
        outs = [(np.empty((world, rows, nth)), np.zeros(hi - lo, dtype=M._capi.INFO_DTYPE)) for _ in range(4)]

        def steps(K):
            pend, last = [], None
            for k in range(K):
                prob.map_and_score_batch_gather_async(seed, lo, hi, theta, rows, atol=1e-2, z0_mode=M.Z0_ZERO, result_area=k % 4)
                pend.append(k % 4)
                if len(pend) > 3:
                    a = pend.pop(0)
                    last = prob.batch_wait_gathered(hi - lo, rows, a, out=outs[a])
            while pend:
                a = pend.pop(0)
                last = prob.batch_wait_gathered(hi - lo, rows, a, out=outs[a])
            return last
        steps(1)
        K = 4
        sync(prob)
        t0 = time.perf_counter()
        g_all, info = steps(K)
        sync(prob)
        dt = tmax(time.perf_counter() - t0) / K
        ok = all_ok(bool(np.all(info["status"] == 0)))
        prob.close()
        out["cfg5_smooth_1e5"] = {"ms_per_step": 1e3 * dt, "sims_per_s": nsims / dt, "nsims": nsims, "N": N, "sims_per_rank": hi - lo,
                                  "converged_on_every_rank": ok}
    except Exception as e:  # noqa: BLE001
        out["cfg5_smooth_1e5"] = {"skipped": f"{type(e).__name__}: {e}"}
    return out


def flat_sharded(cfg, sh):
    """The sharded extras as FLAT scalars of `config` (the driver's record keeps flat scalars of config / roofline only)."""
    mr = sh.get("muse_run", {})
    if "runs" in mr:
        d = mr["runs"]["default"]
        cfg.update({"sharded_muse_iter_us": d["us_per_outer_iteration_30"], "sharded_muse_iter_steady_us": d["us_per_outer_iteration_steady"],
                    "sharded_loop_ran": d["loop_ran"], "sharded_board": mr["board"], "sharded_bit_equal": bool(mr["trajectory_bit_equal_to_unsharded_on_every_rank"]),
                    "sharded_ranks_seen": mr["ranks_seen"], "handshake_device": mr["handshake"]["device_handshake"],
                    "handshake_host": mr["handshake"]["host_handshake"], "handshake_device_wait_us": mr["handshake"]["device_wait_us"],
                    "handshake_host_wait_us": mr["handshake"]["host_wait_us"], "board_setup_ms": mr["board_setup_ms"],
                    "sharded_muse_iter_host_board_us": mr["runs"]["host_board"]["us_per_outer_iteration_30"],
                    "sharded_muse_iter_host_loop_us": mr["runs"]["host_loop"]["us_per_outer_iteration_30"]})
    else:
        cfg["sharded_muse_skipped"] = mr.get("skipped", "not run")
    for name, key in (("cfg4_fd_H", "ms_per_call"), ("cfg5_smooth_1e5", "ms_per_step")):
        r = sh.get(name, {})
        if key in r:
            cfg[f"{name}_sharded_ms"] = r[key]
            cfg[f"{name}_sharded_ok"] = bool(r["converged_on_every_rank"])
        else:
            cfg[f"{name}_sharded_skipped"] = r.get("skipped", "not run")


def flat_single(out):
    """The single-GPU extras as FLAT scalars of `config` / `roofline` (the driver's record keeps flat scalars of those two objects and
    drops `extra`): one number per workload -- the pipelined two-lane step (one lane for the stencil model) --, the muse! iteration,
    the 8-GPU share and the projections."""
    x, cfg = out.get("extra", {}), out["config"]
    for name, tag in (("funnel4_1e4", "funnel4"), ("noise_1e6", "noise_1e6"), ("smooth_1e5", "smooth_1e5")):
        wl = x.get("workloads", {}).get(name, {})
        if "ms_per_step" in wl:
            cfg[f"{tag}_ms"], cfg[f"{tag}_frac"], cfg[f"{tag}_bound"] = wl["ms_per_step"], wl["frac"], wl["bound"]
            cfg[f"{tag}_frac_kernel_time"] = wl["frac_kernel_time"]
    mr = x.get("muse_run", {})
    if "us_per_outer_iteration_30" in mr:
        cfg["muse_iter_us"] = mr["us_per_outer_iteration_30"]
        cfg["muse_iter_steady_us"] = mr["us_per_outer_iteration_steady"]
        cfg["muse_iter_host_loop_us"] = mr["us_per_outer_iteration_30_host_loop"]
        for reg, v in mr.get("by_regime", {}).items():
            cfg[f"muse_iter_{reg}_us"] = v["us_per_outer_iteration"]
        cfg["muse_full_run_ms"] = 1e3 * mr["wall_s"]
    sh = x.get("muse_run_8gpu_share", {})
    if "sharded_loop_shm_1rank" in sh:
        cfg["share_iter_us"] = sh["sharded_loop_shm_1rank"]["us_per_outer_iteration_steady"]
        cfg["share_iter_30_us"] = sh["sharded_loop_shm_1rank"]["us_per_outer_iteration_30"]
        cfg["proj_muse"] = sh.get("projected_speedup_at_8_gpus")
        for reg, v in sh.get("projected_by_regime", {}).items():
            cfg[f"proj_muse_{reg}"] = v["projected_speedup_at_8_gpus"]
        hs = sh.get("board_handshake", {})
        if hs:
            cfg["handshake_device"], cfg["handshake_host"], cfg["sharded_board"] = hs["device_handshake"], hs["host_handshake"], hs["board"]
    sp = x.get("scale_projection", {})
    for name, tag in (("cfg4_fd_H", "cfg4"), ("cfg5_smooth_1e5", "cfg5")):
        if "projected_speedup_at_8_gpus" in sp.get(name, {}):
            cfg[f"proj_{tag}"] = sp[name]["projected_speedup_at_8_gpus"]
            cfg[f"{tag}_whole_ms"], cfg[f"{tag}_share_ms"] = sp[name]["whole_job_ms"], sp[name]["share_ms"]
    if "muse_map_warm_us_per_step" in x:
        cfg["muse_map_warm_us"] = x["muse_map_warm_us_per_step"]
    if "get_H_fd_configs3" in x:
        cfg["get_H_fd_cfg4_ms"] = x["get_H_fd_configs3"]["ms_per_call"]


def user_model_rates(M, device, N=10000, nsims=512):
    """User-supplied models (include/muse_model.h; the closures of SimpleMuseProblem, src/simple.jl:79-95, as a compiled header)
    at the headline shape, pipelined over the result areas and two lanes like the timed loop: the built-in funnel written as a
    header (models/gaussian_funnel.h: the same instructions, so the same speed -- what the seam costs) and the shipped
    non-Gaussian example (models/cubic.h at theta = 0: ~18 L-BFGS iterations per sim, bound by the HBM traffic of the
    L-BFGS history).  Only with the models' libraries already built (build(): ~45 s of hipcc each -- never inside the bench)."""
    from museinference_jl_amd import build as _b
    out = {}
    for name, theta, steps in (("funnel_built_in_same_loop", 1.0, 400), ("gaussian_funnel", 1.0, 400), ("cubic", 0.0, 60)):
        if name == "funnel_built_in_same_loop":   # the reference point: MUSE_MODEL_FUNNEL through this same Python loop
            model = "funnel"
        else:
            header, lib = os.path.join(_b.MODELS_DIR, name + ".h"), _b.model_lib_path(name)
            if _b._stale(lib, _b.SOURCES + _b.HEADERS + [header, os.path.join(_b.INCLUDE_DIR, "muse_model.h")]):
                out[name] = {"skipped": "library not built (run __graft_entry__.build())"}
                continue
            model = M.ElementwiseModel.packaged(name)
        prob = M.HipMuseProblem(None, model=model, ntheta=1, N=N, device=device)
        prob.set_normals_cache(False)   # (every step draws its simulations; the steps of run() use distinct ranges anyway)
        prob.set_concurrency(2)
        AREAS = 4
        outs = [(np.empty((nsims, 1)), np.zeros(nsims, dtype=M._capi.INFO_DTYPE)) for _ in range(AREAS)]

        def run(K):
            for k in range(K):
                if k >= AREAS - 1:
                    prob.batch_wait(nsims, (k + 1) % AREAS, out=outs[(k + 1) % AREAS])
                prob.map_and_score_batch_async(0, k * nsims, (k + 1) * nsims, [theta], atol=1e-2, z0_mode=0, result_area=k % AREAS)
            for k in range(max(0, K - AREAS + 1), K):
                prob.batch_wait(nsims, k % AREAS, out=outs[k % AREAS])

        run(max(8, steps // 10))
        prob.synchronize()
        t0 = time.perf_counter()
        run(steps)
        prob.synchronize()
        dt = (time.perf_counter() - t0) / steps
        info = outs[(steps - 1) % AREAS][1]
        words = 4 * info["hist_words"].astype(np.int64) + 2 * info["iterations"].astype(np.int64) + 1
        out[name] = {"us_per_step": 1e6 * dt, "sims_per_s": nsims / dt, "iterations_mean": float(info["iterations"].mean()),
                     "f_calls_mean": float(info["f_calls"].mean()), "theta": theta,
                     "lbfgs_history_TBps": 8.0 * N * float(words.sum()) / dt / 1e12}
        prob.close()
    out["note"] = ("a model given as a C header with three functions, compiled into an engine library of its own (MUSE_MODEL_USER); "
                   "lbfgs_history_TBps: (4 sum h_k + 2 K + 1) N doubles per sim / step time -- the (dx, dg) history is what the resident "
                   "placement keeps in HBM (meaningful for the many-iteration cubic model, not for the one-iteration funnel)")
    return out


def main_fd(args, M, torch, dist, world, rank, local_rank, sharded, tdev):
    """--workload cfg4_fd_H: BASELINE.json configs[3], get_H! by finite differences (src/muse.jl:407-446), 512 sims x 4 theta.  A step
    is ONE call: the rank's block of the flattened (sim, column) list (muse_fd_jacobian_columns) and, with N > 1, the exchange of
    the column blocks through the engine's shared-memory communicator.  The unit of `value` is one MAP+score problem."""
    w = dict(FD_WORKLOAD)
    if args.nsims > 0:
        w["nsims"] = args.nsims   # development aid (the eight-rank test on one GPU)
    nth, N, seed = w["ntheta"], w["N"], 0
    nunits = w["nsims"] * nth
    prob = M.HipMuseProblem(None, model=w["model"], ntheta=nth, N=N, device=local_rank)
    lo, hi = M.block_partition(0, nunits, world, rank)
    collective, seen = None, 1
    if sharded:
        uid = [M.HipMuseProblem.comm_unique_id("shm", 4096) if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        prob.comm_init(world, rank, uid[0])
        collective, seen = "shm-capi", prob.comm_ranks_seen()

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()
        prob.synchronize()

    for _ in range(max(1, args.warmup)):
        cols, info = fd_call(M, prob, seed, lo, hi, sharded)
    barrier()
    rounds = []
    while True:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            cols, info = fd_call(M, prob, seed, lo, hi, sharded)
        barrier()
        dt = time.perf_counter() - t0
        if sharded:
            tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        rounds.append(dt)
        more = 1 if (sum(rounds) < args.min_seconds and len(rounds) < 100000) else 0
        if sharded:
            flag = torch.tensor([more if rank == 0 else 0], dtype=torch.int32, device=tdev)
            dist.broadcast(flag, src=0)
            more = int(flag.item())
        if not more:
            break
    dt = sum(rounds) / len(rounds)
    # roofline leg: HIP events around every solver launch of a few calls; a call is two launches -- the fiducial MAP with the
    # normals-only elements that fill the cache, then the perturbed problems (the dominant one)
    ncalls = 16
    prob.profile_begin(2 * ncalls + 8)
    for _ in range(ncalls):
        fd_call(M, prob, seed, lo, hi, sharded)
    barrier()
    kernel_ms = prob.profile_end()
    fid_ms, fd_ms = kernel_ms[0::2], kernel_ms[1::2]
    assert np.all(info["status"] == 0), "a MAP solve did not converge in the timed region"
    nprob_job = 2 * nunits + 1                      # what the job needs; every rank repeats the one shared fiducial MAP
    nprob_rank = 2 * (hi - lo)
    # compulsory HBM bytes of the perturbed launch (resident placement): a problem reads its simulation's cached normals n1, n2 and
    # the fiducial MAP it starts from; its MAP is not stored (only the score leaves): 3 words per element
    comp = 3 * 8 * N * nprob_rank
    launch_s = float(fd_ms.mean()) * 1e-3
    pinfo = prob.placement_info()
    roof = {"bound": "hbm", "achieved": comp / launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": comp / launch_s / 1e9 / HBM_PEAK_GBS,
            "traffic": None, "kernel": "map_score_kernel (BATCH_FD)", "placement": "resident" if pinfo["resident"] else "streaming",
            "compulsory_bytes_per_launch": comp, "kernel_ms_mean": float(fd_ms.mean()), "fiducial_launch_ms_mean": float(fid_ms.mean()),
            "launches_timed": int(fd_ms.size),
            "note": "the perturbed problems' launch: per problem the simulation's cached normals (2 words per element) and the fiducial MAP "
                    "(1) are read, one L-BFGS iteration runs on chip, nothing but the score is written; a mixed bound -- 3 words per "
                    "element against ~60 fp64 operations",
            "per_sim": {"f_calls_mean": float(info["f_calls"].mean()), "iterations_mean": float(info["iterations"].mean())}}
    out = {
        "metric": "MC sims/sec (MAP+score)", "value": nprob_job * args.steps / dt, "unit": "sims/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"cfg4_fd_H: BASELINE.json configs[3], get_H! by finite differences (central_fdm(3,1), src/muse.jl:407-446) of the "
                               f"{nth}-theta funnel, N={N}, {w['nsims']} sims: {nprob_job} MAP+score problems per step (one 'sim' of the metric = "
                               f"one MAP+score problem), {nprob_rank + 1} on this rank",
                   "theta": w["theta"], "step": w["step"],
                   "parallelism": f"the flattened (sim, column) list in {world} contiguous block(s) (src/muse.jl:327-333 across ranks), one exchange "
                                  "of the column blocks per call" + (f" ({collective}, {seen} ranks seen)" if collective else "")},
        "timed_rounds": len(rounds), "timed_seconds": sum(rounds), "roofline": roof,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(w["model"], N, w["theta"], seed, cpu_seconds=15.0)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    prob.close()
    if sharded:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="funnel_1e4", choices=sorted(WORKLOADS))
    ap.add_argument("--placement", type=int, default=-1, help="-1 auto, 0 streaming, 1 resident")
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"],
                    help="with --gpus N > 1: strong (default) = the step's nsims shared by the ranks, weak = nsims per rank")
    ap.add_argument("--split", type=int, default=-1, help="workgroups per element (-1: by the rank's element count)")
    ap.add_argument("--min-seconds", type=float, default=0.5,
                    help="repeat the timed region (EXACTLY --steps steps between barriers, every time) until this much "
                         "time has been measured; ms_per_step is the mean over the repetitions")
    ap.add_argument("--nsims", type=int, default=0, help="development aid: sims per step instead of the workload's own "
                    "(e.g. 64 = one rank's share of the strongly scaled 8-GPU step, on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed muse!/get_H! rates (N = 1) / the sharded muse! loop, configs[3] "
                    "and configs[4] beside the headline (N > 1)")
    ap.add_argument("--small", action="store_true", help="development aid: the N > 1 extras at sizes that eight processes can share ONE GPU with")
    return ap.parse_args(argv)


def launch_ranks(script, argv, nranks, extra_env=None, grace_s=20.0):
    """Start `nranks` fresh processes of `script` (one rank per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, HIP_VISIBLE_DEVICES untouched), relay rank 0's JSON line to stdout and everything else to stderr, and
    return the worst exit code.  The calling process has not touched the GPU (it never will: it only waits), so nothing
    that has initialised HIP is ever re-executed.  This is what `python bench.py --gpus N` does when it was not started
    by torch.distributed.run; a rank that fails takes the others down after `grace_s` seconds (they would otherwise sit
    in a rendezvous or a collective until its time-out)."""
    import socket
    import subprocess
    import threading
    # a free rendezvous port on the loopback interface, BELOW the ephemeral range: a port handed out by bind(0) goes back to the pool when
    # the probe socket closes, and one of the ranks' own outgoing connections (gloo opens many) can be given it before rank 0's store
    # listens on it -- EADDRINUSE, seen once in a while with eight ranks
    import random
    port = None
    for _ in range(64):
        cand = random.randint(15000, 29999)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", cand))
                port = cand
                break
            except OSError:
                continue
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    procs, lines0 = [], []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=subprocess.PIPE, text=True))

    def pump(r, p):
        for line in p.stdout:
            if r == 0 and line.startswith("{"):
                lines0.append(line)
            else:  # RCCL banners and the like
                sys.stderr.write(f"[rank {r}] {line}")
    threads = [threading.Thread(target=pump, args=(r, p), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    worst, t_fail = 0, None
    while any(p.poll() is None for p in procs):
        for p in procs:
            rc = p.poll()
            if rc is not None and rc != 0 and t_fail is None:
                worst, t_fail = rc, time.monotonic()
        if t_fail is not None and time.monotonic() - t_fail > grace_s:
            for p in procs:  # exactly the processes started above
                if p.poll() is None:
                    p.terminate()
            t_fail = float("inf")
        time.sleep(0.05)
    for t in threads:
        t.join(timeout=5)
    for p in procs:
        rc = p.returncode
        if rc != 0 and (worst == 0 or abs(rc) > abs(worst)):
            worst = rc
    if worst == 0 and not lines0:
        sys.stderr.write("[bench] rank 0 printed no result line\n")
        worst = 1
    if lines0:
        sys.stdout.write(lines0[-1])
        sys.stdout.flush()
    return worst if worst >= 0 else 128 - worst  # a signal's number, shell style


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: become the launcher (before torch or HIP is imported)
        sys.exit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    import torch
    import museinference_jl_amd as M

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or let bench.py do it: "
                         "unset WORLD_SIZE)")
    dist = None
    sharded = world > 1 or os.environ.get("MUSE_BENCH_FORCE_DIST") == "1"  # the env var exercises the N>1 path on one GPU
    # MUSE_BENCH_BACKEND=gloo (development aid): several ranks on ONE GPU, gloo collectives on host tensors -- exercises the
    # partition / padding / gather logic of the N > 1 path on a one-GPU box (RCCL refuses two ranks on one device)
    backend = os.environ.get("MUSE_BENCH_BACKEND", "nccl")
    tdev = "cuda" if backend == "nccl" else "cpu"
    if backend != "nccl":
        local_rank = 0
    if sharded:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.workload == "cfg4_fd_H":   # configs[3]: a step is a get_H! call, not a map
        return main_fd(args, M, torch, dist, world, rank, local_rank, sharded, tdev)
    model, N, nth, theta, nsims = WORKLOADS[args.workload]
    if args.nsims > 0:
        nsims = args.nsims
    seed = 0
    prob = M.HipMuseProblem(None, model=model, ntheta=nth, N=N, device=local_rank)
    # The timed steps repeat ONE map (the same simulations, cold start) to time sample -> MAP -> score: the engine's cache of
    # the standard normals of repeated simulations would turn every step after the second into a load instead of a draw --
    # work skipped inside the timed region.  Off: every step runs the generator.
    prob.set_normals_cache(False)
    if args.placement >= 0:
        prob.set_placement(args.placement)
    # consecutive steps on alternating lanes (streams): a launch starts on the compute units the previous one has left
    # (measured, one GPU: funnel_1e4 50.2 -> 46.1 us per step, funnel4_1e4 59.7 -> 54.0, noise_1e6 1.377 -> 1.268 ms -- the
    # HBM-only tail of one launch runs beside the generator-only head of the next; the stencil model's 16-member clusters, whose
    # launches would have to share the compute units, lose 4 %: one lane)
    lanes = int(os.environ.get("MUSE_BENCH_LANES", "1" if model == "smooth" else "2"))
    if lanes > 1:
        prob.set_concurrency(lanes)
    scaling = args.scaling or ("strong" if world > 1 else "weak")
    if scaling == "strong":
        # the step's nsims sims are shared by the ranks: contiguous blocks (the reference's pmap over a worker pool
        # splits the same list, src/muse.jl:169 with src/util.jl:74-83); the gathered block is padded to the largest
        sim_lo, sim_hi = M.block_partition(0, nsims, world, rank)
        rows = -(-nsims // world)
        total_sims = nsims
    else:
        sim_lo, sim_hi = rank * nsims, (rank + 1) * nsims  # this rank's block of the global nsims*world map
        rows = nsims
        total_sims = nsims * world
    nlocal = sim_hi - sim_lo
    cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
    # The exchange of the per-rank score blocks (N > 1).  Both transports of the engine are measured in the same run, one
    # after the other, and both go into the line (`transports`); `value` is the better one.  "shm": the ranks of this
    # bench share a node (the launch contract), the blocks travel host to host through the engine's shared-memory
    # segment; "rccl": device-side RCCL all-gather over xGMI on a second stream.  MUSE_BENCH_TRANSPORT=shm|rccl: only one.
    want = os.environ.get("MUSE_BENCH_TRANSPORT", "both")
    if want not in ("shm", "rccl", "both"):
        raise SystemExit("MUSE_BENCH_TRANSPORT must be shm, rccl or both")
    transports = [None] if not sharded else (["shm", "rccl"] if want == "both" else [want])
    same_node = True
    if sharded:
        same_node = M.ranks_share_node(dist)

    AREAS = 4
    state = {"capi": False, "collective": None, "gather_buf": None, "nmaps": 1}
    theta_rep = np.tile(np.asarray(theta, dtype=np.float64), (M._capi.MAX_MAPS, 1))
    host_t = [0.0, 0.0]  # host seconds spent enqueueing / waiting+collecting (reported under "host_us_per_step")

    def run_steps(K, collect=None):
        """K steps, software-pipelined: batch k is enqueued before batch k-1's results are awaited,
        so the GPU never idles on the host; with >1 GPU the all-gather of step k-1 overlaps batch k.
        With maps_per_launch = G > 1 (a rank's share of the step is smaller than the GPU) one launch carries G consecutive
        steps -- G independent maps resident at once -- and one exchange serves all of them."""
        pending = []
        capi, G = state["capi"], state["nmaps"]
        k = j = 0
        while k < K:
            m = min(G, K - k)
            t_enq0 = time.perf_counter()
            if G > 1:
                if capi:
                    n = prob.map_and_score_multi_gather_async(seed, sim_lo, sim_hi, theta_rep[:m], rows, atol=1e-2,
                                                              z0_mode=M.Z0_ZERO, result_area=j % AREAS)
                else:
                    n = prob.map_and_score_multi_async(seed, sim_lo, sim_hi, theta_rep[:m], atol=1e-2, z0_mode=M.Z0_ZERO,
                                                       result_area=j % AREAS)
            elif capi:
                n = prob.map_and_score_batch_gather_async(seed, sim_lo, sim_hi, theta, rows, atol=1e-2,
                                                          z0_mode=M.Z0_ZERO, result_area=j % AREAS)
            else:
                n = prob.map_and_score_batch_async(seed, sim_lo, sim_hi, theta, atol=1e-2, z0_mode=M.Z0_ZERO,
                                                   result_area=j % AREAS)
            host_t[0] += time.perf_counter() - t_enq0
            pending.append((j % AREAS, n, m))
            if len(pending) > AREAS - 1:
                t_w0 = time.perf_counter()
                finish(pending.pop(0), collect)
                host_t[1] += time.perf_counter() - t_w0
            k += m
            j += 1
        while pending:
            finish(pending.pop(0), collect)

    # result buffers of the host loop, one pair per result area (a step's results are looked at before its area comes round)
    outs = {}

    def out_for(area, n, m):
        key = (area, n, m, state["capi"])
        if key not in outs:
            shape = (world, m * rows, nth) if state["capi"] else (n, nth)
            outs[key] = (np.empty(shape), np.zeros(n, dtype=M._capi.INFO_DTYPE))
        return outs[key]

    def finish(item, collect):
        area, n, m = item
        if state["capi"]:
            # [world, m * rows, nth]: all ranks' scores of the launch's m maps
            g_all, info = prob.batch_wait_gathered(n, m * rows, area, out=out_for(area, n, m))
            g = g_all[rank].reshape(m, rows, nth)[:, :nlocal]
        else:
            g, info = prob.batch_wait(n, area, out=out_for(area, n, m))
            if state["collective"] == "torch":
                pad = np.zeros(state["nmaps"] * rows * nth)
                pad[: g.size] = g.reshape(-1)
                dist.all_gather(state["gather_buf"], torch.from_numpy(pad).to(tdev))
        if collect is not None:
            collect.append((g, info))

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()
        prob.synchronize()

    def setup_exchange(transport):
        """Element split and communicator for one transport; returns (split, collective, ranks_seen) or a reason it
        cannot run (str) -- the same answer on every rank."""
        # fewer elements than compute units: the element itself is the remaining parallel axis (src/muse.jl:327-333
        # chooses the longer axis; here: `split` workgroups per element).  The same split on every rank.
        # Steps of this bench are independent maps (the get_J!-style pass), so a rank whose share of a step is smaller than
        # its GPU keeps the GPU full with SEVERAL steps resident at once: G of them in one launch, one exchange for all
        # (MUSE_BENCH_MAPS: 1 = one step per launch, with the element split below instead -- what a muse! iteration,
        # whose next step depends on this one's scores, has to do).
        G = 1
        resident_n = model != "smooth" and N <= M.load_library().muse_max_resident_n()   # one workgroup per element and CU
        if sharded and args.split < 0 and resident_n:
            # as many maps as make the launch the shape of the 1-GPU step: two problems per compute unit
            # (from `rows`, the largest block: the same count on every rank -- with 512 sims on 3 ranks the blocks are 171, 171, 170)
            G = max(1, min(M._capi.MAX_MAPS, (2 * cus) // max(1, rows))) if os.environ.get("MUSE_BENCH_MAPS") is None \
                else max(1, min(M._capi.MAX_MAPS, int(os.environ["MUSE_BENCH_MAPS"])))
        state["nmaps"] = G
        split = args.split
        if split < 0:
            split = 1
            # (resident placements only: above muse_max_resident_n, and for the stencil model, the cluster size is a
            # function of N alone already; 8 never paid, tools/split_bench.py)
            splittable = model != "smooth" and 512 < N <= M.load_library().muse_max_resident_n()
            # with a collective in flight beside the solver (N > 1) the clusters fill half of the CUs at most: the RCCL
            # kernel of the previous step and a cluster launch that needs every CU would otherwise wait for each other
            room = 4 if transport == "rccl" else 2
            while G == 1 and splittable and split < 4 and room * split * rows <= cus:
                split *= 2
        prob.set_element_split(split if split > 1 else 0)
        if transport is None:
            return split, None, 1
        if transport == "shm" and not same_node:
            return "the ranks do not share a node (boot id / /dev/shm probe)"
        if transport == "rccl" and backend != "nccl":
            return "gloo test mode: RCCL refuses two ranks on one device"
        # Preferred: the engine's own communicator (shm: host-to-host blocks in a shared segment; rccl: scores stay on the
        # device, the all-gather runs on a second stream from C).  Fallback, agreed on by all ranks: torch.distributed's
        # all_gather.
        ok, seen = 1, 0
        try:
            uid = [M.HipMuseProblem.comm_unique_id(transport) if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            prob.comm_init(world, rank, uid[0])
            seen = prob.comm_ranks_seen()
        except Exception as e:  # noqa: BLE001 -- any failure means "use the fallback", on every rank
            print(f"[bench rank {rank}] engine communicator ({transport}) unavailable ({e}); using torch.distributed", file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        use_capi = int(flag.item()) == 1 and os.environ.get("MUSE_BENCH_COLLECTIVE") != "torch"
        if not use_capi:
            prob.comm_destroy()
            state["gather_buf"] = [torch.empty(state["nmaps"] * rows * nth, dtype=torch.float64, device=tdev) for _ in range(world)]
            return split, "torch", dist.get_world_size()
        return split, f"{transport}-capi", seen

    def measure(transport):
        """Warm-up, the timed region (EXACTLY --steps steps between barriers, repeated until --min-seconds), and the
        roofline leg (HIP events around every solver launch) for one transport."""
        got = setup_exchange(transport)
        if isinstance(got, str):
            return {"skipped": got}
        split, collective, seen = got
        state["collective"] = collective
        state["capi"] = collective is not None and collective.endswith("-capi")
        prob.set_timing(False)
        run_steps(args.warmup)
        barrier()
        rounds, results = [], []
        while True:
            results.clear()
            host_t[0] = host_t[1] = 0.0
            t0 = time.perf_counter()
            run_steps(args.steps, results)
            barrier()
            dt = time.perf_counter() - t0
            if sharded:
                tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dt = float(tmax.item())
            rounds.append(dt)
            if os.environ.get("MUSE_BENCH_DEBUG_ROUNDS") and rank == 0:
                print(f"round {len(rounds)}: {1e6 * dt / args.steps:.1f} us/step", file=sys.stderr)
            more = 1 if (sum(rounds) < args.min_seconds and len(rounds) < 100000) else 0
            if sharded:  # every rank runs the same number of repetitions (rank 0 decides)
                flag = torch.tensor([more if rank == 0 else 0], dtype=torch.int32, device=tdev)
                dist.broadcast(flag, src=0)
                more = int(flag.item())
            if not more:
                break
        dt = sum(rounds) / len(rounds)
        host_us = {"enqueue": 1e6 * host_t[0] / args.steps, "wait_and_collect": 1e6 * host_t[1] / args.steps}
        # roofline leg: the same steps again with a HIP event pair around every solver launch, recorded on the stream the
        # kernel is launched on
        nprof = min(max(args.steps, 64), 256)
        prob.profile_begin(nprof + 8)
        run_steps(nprof)
        barrier()
        kernel_ms = prob.profile_end()
        try:
            clock_hz = prob.profile_clock_hz()   # in-kernel: d(s_memtime) / d(s_memrealtime) x 100 MHz of a profiled launch
        except M.MuseError:
            clock_hz = None
        g, info = results[-1]
        assert np.all(info["status"] == 0), "a MAP solve did not converge in the timed region"
        pinfo = prob.placement_info()
        if state["capi"]:
            prob.comm_destroy()
        state["capi"] = False
        return {"dt": dt, "rounds": rounds, "host_us": host_us, "kernel_ms": kernel_ms, "info": info, "split": split,
                "maps_per_launch": state["nmaps"],
                "collective": collective, "ranks_seen": seen, "pinfo": pinfo, "value": total_sims * args.steps / dt,
                "clock_hz": clock_hz, "launches": -(-args.steps // state["nmaps"])}

    # The host loop must keep three launches ahead of a ~50 us kernel: a full collection of the interpreter's cyclic GC
    # (torch alone brings more than a million tracked objects; measured 40-60 ms, once, a few hundred steps into a run)
    # drains the pipeline.  Everything alive now is setup: collect it once and move it out of the collector's sight.
    import gc
    gc.collect()
    gc.freeze()
    def build_line(measured):
        """The result line from what has been measured (all transports, or -- from the watchdog -- the ones that finished)."""
        ran = {t: m for t, m in measured.items() if "skipped" not in m}
        if not ran:
            raise SystemExit(f"no transport could run: {measured}")
        best_t = max(ran, key=lambda t: ran[t]["value"])
        best = ran[best_t]
        dt, rounds, host_us, kernel_ms, info, split, collective, pinfo = (best[k] for k in (
            "dt", "rounds", "host_us", "kernel_ms", "info", "split", "collective", "pinfo"))

        prow_ok = world == 1 and split == 1 and not sharded
        launch_s = dt / best["launches"] if world == 1 else float(kernel_ms.mean()) * 1e-3
        primary = roofline_object(args.workload, model, N, info, pinfo, kernel_ms, launch_s, best["clock_hz"], lanes,
                                  1e3 * dt / best["launches"], prow_ok)

        out = {
            "metric": "MC sims/sec (MAP+score)",
            "value": total_sims * args.steps / dt,
            "unit": "sims/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: Neal's funnel family model={model}, N={N}-dim z, {nth}-dim theta, "
                                   f"nsims={total_sims} per step ({nlocal} on this rank), cold start z0=0, atol=1e-2",
                       "theta": theta, "sims_per_step_total": total_sims, "element_split": split,
                       "pipelining": f"maps_per_launch={best['maps_per_launch']}: that many consecutive (independent) steps share one "
                                     "launch and one exchange",
                       "parallelism": f"sims sharded over {world} GPU(s) ({scaling} scaling), one all-gather of scores per step"
                                      + (f" ({collective})" if collective else ""),
                       "scaling_note": ("the timed steps are INDEPENDENT maps (a get_J!-style pass repeated): with N > 1 a launch carries "
                                        "maps_per_launch of them, so `value` is the strong scaling of independent maps per launch -- not of "
                                        "one dependent map.  A muse! iteration (src/muse.jl:159-232) is one dependent map per launch: its "
                                        "8-GPU share is measured on one GPU in extra.muse_run_8gpu_share (N = 1 runs), projected speed-up "
                                        "2-3x, bounded by one problem's latency plus the exchange")},
            "timed_rounds": len(rounds), "timed_seconds": sum(rounds),
            "ms_per_step_min_round": 1e3 * min(rounds) / args.steps, "ms_per_step_max_round": 1e3 * max(rounds) / args.steps,
            "roofline": primary,
            "host_us_per_step": host_us,
        }
        if sharded:
            # both exchanges of the same run, side by side: did RCCL see N ranks, and what did each cost
            out["transport"] = best_t
            out["transports"] = {
                t: (m if "skipped" in m else
                    {"value": m["value"], "ms_per_step": 1e3 * m["dt"] / args.steps, "collective": m["collective"],
                     "ranks_seen": m["ranks_seen"], "element_split": m["split"],
                     "pipelining": f"maps_per_launch={m['maps_per_launch']}",
                     "kernel_ms_mean": float(m["kernel_ms"].mean()), "host_us_per_step": m["host_us"],
                     "timed_rounds": len(m["rounds"])})
                for t, m in measured.items()}
        return out

    # A transport whose collective never completes (RCCL over a broken link, a peer that died) must not cost the line: every
    # rank arms a watchdog for the transports after the first; when it fires, rank 0 prints the line of what finished -- the
    # hung transport marked failed ("transport_failed") -- and every rank leaves with status 3 (os._exit: the main thread is
    # inside a C call; the status is non-zero because a collective hung with GPU work in flight).
    import threading
    measured = {}
    deadline = float(os.environ.get("MUSE_BENCH_TRANSPORT_DEADLINE_S", "240"))
    for idx, t in enumerate(transports):
        timer = None
        if idx > 0:
            def fire(t=t):
                print(f"[bench rank {rank}] transport {t} did not finish within {deadline:.0f} s: reporting without it", file=sys.stderr)
                if rank == 0:
                    line = build_line(dict(measured, **{t: {"skipped": f"did not finish within {deadline:.0f} s (watchdog)"}}))
                    line["transport_failed"] = t
                    sys.stdout.write(json.dumps(line) + "\n")
                    sys.stdout.flush()
                # non-zero: these ranks have GPU work of a hung collective in flight -- the launcher (and CI) must see a failed
                # run, not a success with a "skipped" string inside the line
                os._exit(3)
            timer = threading.Timer(deadline, fire)
            timer.daemon = True
            timer.start()
        measured[t] = measure(t)
        if timer is not None:
            timer.cancel()
    out = build_line(measured)
    best_t = out.get("transport")
    if rank == 0 and world == 1 and not sharded and not args.no_extra:
        out["extra"] = extra_rates(M, prob, model, N, nth, theta, nsims, seed, local_rank)
        if args.workload == "funnel_1e4":
            # the other BASELINE workloads, briefly, so that they are in the driver's line too (each <= ~0.3 s of timed steps)
            out["extra"]["workloads"] = {}
            for name in ("funnel4_1e4", "noise_1e6", "smooth_1e5"):
                try:
                    out["extra"]["workloads"][name] = quick_workload(M, name, local_rank)
                except Exception as e:  # noqa: BLE001 -- an extra: never at the cost of the line
                    out["extra"]["workloads"][name] = {"skipped": f"{type(e).__name__}: {e}"}
            out["extra"]["scale_projection"] = scale_projection(M, local_rank)
        if args.workload == "funnel_1e4":
            try:
                out["extra"]["user_model"] = user_model_rates(M, local_rank)
            except Exception as e:  # an extra: never at the cost of the line
                out["extra"]["user_model"] = {"skipped": f"{type(e).__name__}: {e}"}
        flat_single(out)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model, N, theta, seed)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if sharded and not args.no_extra and args.workload == "funnel_1e4":
        # beside the headline's independent maps: the dependent path and the two workloads BASELINE.json puts on 8 GPUs, over THESE ranks
        prob.close()
        # (a collective of the extras that never completes -- a peer that died, a board wait -- must not cost the line: the same
        #  watchdog as for a hung transport; it prints the headline without the extras and leaves with status 3)
        def fire_extras():
            print(f"[bench rank {rank}] the sharded extras did not finish within {deadline:.0f} s: reporting without them", file=sys.stderr)
            if rank == 0:
                out["config"]["sharded_muse_skipped"] = f"did not finish within {deadline:.0f} s (watchdog)"
                sys.stdout.write(json.dumps(out) + "\n")
                sys.stdout.flush()
            os._exit(3)
        timer = threading.Timer(deadline, fire_extras)
        timer.daemon = True
        timer.start()
        sh = sharded_extras(M, torch, dist, args, world, rank, local_rank, tdev)
        timer.cancel()
        out["sharded"] = sh
        flat_sharded(out["config"], sh)
    if sharded:
        prob.close()
        dist.destroy_process_group()
    if rank == 0:  # last: RCCL writes its own banner to stdout while the communicator is set up and torn down
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
