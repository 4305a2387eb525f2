#!/usr/bin/env python3
"""bench.py -- MC sims/sec (MAP+score) of the MUSE inner loop on MI355X.

One "step" = one pass of the hot path over one batch: for every one of the 512 simulations of the
batch, sample (x, z) ~ P(x, z | θ), find the latent MAP ẑ by L-BFGS/HagerZhang from a cold start
ẑ₀ = 0 to ||∇z||∞ <= 1e-2, and evaluate the score ∇θ logP(x, ẑ | θ) -- the get_J!/muse! map body of
the reference (src/muse.jl:169-176, :508-525), one launch per batch.  Workload at N=1 GPU:
BASELINE.json configs[1], Neal's funnel, 10^4-dim z, 1-dim θ (θ = 1), nsims = 512, fp64, synthetic
(Philox) data.  With --gpus N every rank runs its own 512-sim block of a 512*N-sim map (weak scaling;
sims are independent) and the per-rank score blocks are exchanged with one all-gather per step (RCCL).

Prints ONE JSON line on rank 0 (see the driver contract), with two extra objects:
  roofline      algorithmic HBM bytes of the solver kernel per launch / its mean launch duration
                (HIP events on the launch stream), against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (oracle/, a restatement of the reference: "port") timed on this box's
                host cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
WORKLOADS = {
    # name: (model, N, ntheta, theta, nsims)
    "funnel_1e4": ("funnel", 10000, 1, [1.0], 512),        # BASELINE.json configs[1] (headline)
    "funnel_512": ("funnel", 512, 1, [1.0], 32),           # configs[0]
    "noise_1e6": ("noise", 1000000, 1, [0.5], 128),        # configs[2]
    "funnel4_1e4": ("funnel", 10000, 4, [1.0] * 4, 512),   # configs[3] (s/J pass)
    "smooth_1e5": ("smooth", 100000, 8, [1.0] * 8, 128),   # configs[4], per-GPU share
}


def algorithmic_bytes(info, N):
    """SURVEY.md §8(d3) / BASELINE.md §3: words = 1 + 5E + Σ_k(4 h_k + 4) + 2 per sim, 8 B words."""
    E = info["f_calls"].astype(np.int64)
    K = info["iterations"].astype(np.int64)
    H = info["hist_words"].astype(np.int64)
    words = 1 + 5 * E + 4 * H + 4 * K + 2
    return int(8 * N * words.sum())


def measured_traffic(workload):
    """HBM bytes per solver launch from the committed rocprofv3 PMC passes (profiles/r01_summary.csv:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB,
    the factor 2 being the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md §HBM).  None if not profiled."""
    path = os.path.join(ROOT, "profiles", "r01_summary.csv")
    try:
        import csv
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["workload"] == workload:
                    return float(row["hbm_traffic_MB"]) * 1e6
    except OSError:
        pass
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
    return {
        "value": nall / tall, "unit": "sims/s", "cores": cores, "kind": "port",
        "sample": f"{nall} sims of the same workload (oracle/muse_oracle.c, gcc -O3, OpenMP over sims, "
                  f"{cores} threads, {tall:.2f} s wall); 1 thread: {1.0 / t1:.1f} sims/s on {n1} sims",
        "value_1thread": 1.0 / t1,
    }


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
    out["muse_run"] = {"wall_s": dt, "outer_iterations": len(res.history), "theta": [float(t) for t in res.theta],
                       "sigma": [float(t) for t in np.sqrt(np.diag(np.atleast_2d(res.Sigma)))],
                       "us_per_outer_iteration_30": 1e6 * dt30 / max(1, len(res30.history)),
                       "note": "muse(prob, theta0=1; nsims, get_covariance=True): outer iterations (native muse_run) + get_J! + get_H!, "
                               "host algebra included; us_per_outer_iteration_30 from a 30-iteration run"}
    prob.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="funnel_1e4", choices=sorted(WORKLOADS))
    ap.add_argument("--placement", type=int, default=-1, help="-1 auto, 0 streaming, 1 resident")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed muse!/get_H! rates")
    args = ap.parse_args()

    import torch
    import museinference_jl_amd as M

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dist = None
    sharded = world > 1 or os.environ.get("MUSE_BENCH_FORCE_DIST") == "1"  # the env var exercises the N>1 path on one GPU
    if sharded:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world)

    model, N, nth, theta, nsims = WORKLOADS[args.workload]
    seed = 0
    prob = M.HipMuseProblem(None, model=model, ntheta=nth, N=N, device=local_rank)
    if args.placement >= 0:
        prob.set_placement(args.placement)
    sim0 = rank * nsims  # this rank's block of the global map
    gather_buf = None
    collective = None
    if sharded:
        # Preferred: the engine's own RCCL communicator (scores stay on the device, the all-gather runs on a
        # second stream from C).  Fallback, agreed on by all ranks: torch.distributed's all_gather.
        ok = 1
        try:
            uid = [M.HipMuseProblem.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            prob.comm_init(world, rank, uid[0])
        except Exception as e:  # noqa: BLE001 -- any failure means "use the fallback", on every rank
            print(f"[bench rank {rank}] engine RCCL communicator unavailable ({e}); using torch.distributed", file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        collective = "rccl-capi" if int(flag.item()) == 1 and os.environ.get("MUSE_BENCH_COLLECTIVE") != "torch" else "torch"
        if collective == "torch":
            gather_buf = [torch.empty(nsims * nth, dtype=torch.float64, device="cuda") for _ in range(world)]

    AREAS = 4
    host_t = [0.0, 0.0]  # host seconds spent enqueueing / waiting+collecting (reported under "host_us_per_step")

    def run_steps(K, collect=None):
        """K steps, software-pipelined: batch k is enqueued before batch k-1's results are awaited,
        so the GPU never idles on the host; with >1 GPU the all-gather of step k-1 overlaps batch k."""
        pending = []
        for k in range(K):
            t_enq0 = time.perf_counter()
            if collective == "rccl-capi":
                n = prob.map_and_score_batch_gather_async(seed, sim0, sim0 + nsims, theta, nsims, atol=1e-2,
                                                          z0_mode=M.Z0_ZERO, result_area=k % AREAS)
            else:
                n = prob.map_and_score_batch_async(seed, sim0, sim0 + nsims, theta, atol=1e-2, z0_mode=M.Z0_ZERO,
                                                   result_area=k % AREAS)
            host_t[0] += time.perf_counter() - t_enq0
            pending.append((k % AREAS, n))
            if len(pending) > AREAS - 1:
                t_w0 = time.perf_counter()
                finish(pending.pop(0), collect)
                host_t[1] += time.perf_counter() - t_w0
        while pending:
            finish(pending.pop(0), collect)

    def finish(item, collect):
        area, n = item
        if collective == "rccl-capi":
            g_all, info = prob.batch_wait_gathered(n, nsims, area)  # [world, nsims, nth]: every rank holds all scores
            g = g_all[rank]
        else:
            g, info = prob.batch_wait(n, area)
            if collective == "torch":
                t = torch.from_numpy(np.ascontiguousarray(g.reshape(-1))).cuda()
                dist.all_gather(gather_buf, t)
        if collect is not None:
            collect.append((g, info))

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()
        prob.synchronize()

    # timed region: EXACTLY `steps` steps, no per-launch timing events (pure throughput)
    prob.set_timing(False)
    run_steps(args.warmup)
    barrier()
    results = []
    host_t[0] = host_t[1] = 0.0
    t0 = time.perf_counter()
    run_steps(args.steps, results)
    barrier()
    dt = time.perf_counter() - t0
    host_us = {"enqueue": 1e6 * host_t[0] / args.steps, "wait_and_collect": 1e6 * host_t[1] / args.steps}
    if sharded:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    # roofline leg: the same steps again with a HIP event pair around every solver launch, recorded on
    # the stream the kernel is launched on
    nprof = min(args.steps, 256)
    prob.profile_begin(nprof + 8)
    run_steps(nprof)
    barrier()
    kernel_ms = prob.profile_end()

    g, info = results[-1]
    assert np.all(info["status"] == 0), "a MAP solve did not converge in the timed region"
    alg_bytes = algorithmic_bytes(info, N)
    mean_kernel_s = float(kernel_ms.mean()) * 1e-3
    achieved = alg_bytes / mean_kernel_s / 1e9

    out = {
        "metric": "MC sims/sec (MAP+score)",
        "value": world * nsims * args.steps / dt,
        "unit": "sims/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: Neal's funnel family model={model}, N={N}-dim z, {nth}-dim theta, "
                               f"nsims={nsims} per GPU per step, cold start z0=0, atol=1e-2",
                   "theta": theta, "sims_per_step_total": world * nsims,
                   "parallelism": f"sims sharded over {world} GPU(s), one all-gather of scores per step"
                                  + (f" ({collective})" if collective else "")},
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args.workload),
            "kernel": "map_score_kernel", "kernel_ms_mean": 1e3 * mean_kernel_s,
            "kernel_ms_min": float(kernel_ms.min()), "launches_timed": int(kernel_ms.size),
            "algorithmic_bytes_per_launch": alg_bytes,
            "note": "achieved = algorithmic bytes (SURVEY 8.d3 accounting) / kernel time; at N <= 10^4 the accounted vectors live in "
                    "registers/LDS, so achieved exceeds the HBM peak and `traffic` (PMC, profiles/) is what HBM really moved; "
                    "the resident kernel is VALU-issue-bound (DESIGN.md 6)",
            "per_sim": {"f_calls_mean": float(info["f_calls"].mean()), "iterations_mean": float(info["iterations"].mean()),
                        "hist_pairs_mean": float(info["hist_words"].mean())},
        },
        "kernel_sims_per_s": nsims / mean_kernel_s,
        "host_us_per_step": host_us,
    }
    if rank == 0 and world == 1 and not sharded and not args.no_extra:
        out["extra"] = extra_rates(M, prob, model, N, nth, theta, nsims, seed, local_rank)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model, N, theta, seed)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out))
    if sharded:
        prob.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
