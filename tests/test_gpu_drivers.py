"""-m gpu: the drivers end to end on the HIP path, golden fixtures, full-size properties at
BASELINE.json's sizes, error behaviour of the C ABI, and the RCCL exchange with one rank."""
import os

import numpy as np
import pytest

from oracle_problem import OracleBatchedProblem

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_golden_fixtures(gpu, M):
    d = np.load(os.path.join(HERE, "golden", "per_sim.npz"))
    keys = sorted({k.rsplit("_", 1)[0] for k in d.files})
    for key in keys:
        model = "funnel" if key.startswith("funnel") else ("noise" if key.startswith("noise") else "smooth")
        N, seed = int(key.split("_N")[1].split("_")[0]), int(key.split("_s")[1])
        th = d[key + "_theta"]
        if model == "smooth" and N < 5:
            continue
        prob = M.HipMuseProblem(None, model=model, ntheta=th.size, N=N)
        x, z = prob.sample_x_z(M.SimRng(seed, 5), th)
        assert np.array_equal(x, d[key + "_x"]) and np.array_equal(z, d[key + "_z"])   # bit-exact sampler
        zh, info = prob.zhat_at_theta(x, np.zeros(N), th, 1e-2)
        assert (info["iterations"], info["f_calls"]) == tuple(d[key + "_iters"][:2])
        np.testing.assert_allclose(zh, d[key + "_zhat"], rtol=0, atol=1e-9)
        zh6, info6 = prob.zhat_at_theta(x, np.zeros(N), th, 1e-6)
        assert (info6["iterations"], info6["f_calls"]) == tuple(d[key + "_iters"][2:])
        np.testing.assert_allclose(zh6, d[key + "_zhat6"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(prob.grad_theta_logLike(x, d[key + "_zhat"], th), d[key + "_score"], rtol=1e-12)
        f, g = prob.logLike_and_grad_z_logLike(x, 0.5 * z, th)
        np.testing.assert_allclose(f, d[key + "_logLike"][0], rtol=1e-12)
        np.testing.assert_allclose(g, d[key + "_gradz"], rtol=1e-13, atol=1e-13)
        _, zf = None, None
        Hs, _ = prob.fd_jacobian_batch(seed, 5, 6, th, np.full(th.size, 0.05), atol=1e-2, fid_mode=0, fid_sim=7)
        np.testing.assert_allclose(Hs[0], d[key + "_H"], rtol=1e-8, atol=1e-8 * np.abs(d[key + "_H"]).max())
        prob.close()


def test_muse_end_to_end_matches_oracle_driven_run(gpu, M, O):
    """muse! + get_J! + get_H! on the GPU against the same drivers on the oracle (rtol 1e-8 on theta-hat,
    J, H, Sigma): identical random streams, identical L-BFGS paths."""
    d = np.load(os.path.join(HERE, "golden", "muse_trajectory.npz"))
    prob = M.HipMuseProblem(d["x"], model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, 3.0))
    res = M.muse(prob, [1.0], rng=42, nsims=32, get_covariance=True)
    np.testing.assert_allclose(np.array([h["θ"] for h in res.history]), d["thetas"], rtol=1e-9)
    np.testing.assert_allclose(res.theta, d["theta"], rtol=1e-8)
    np.testing.assert_allclose(np.array(res.gs), d["gs"], rtol=1e-10)
    np.testing.assert_allclose(res.J, d["J"], rtol=1e-9)
    np.testing.assert_allclose(res.H, d["H"], rtol=1e-7)
    np.testing.assert_allclose(res.Sigma, d["Sigma"], rtol=1e-7)
    assert abs(res.dist.μ) / res.dist.σ < 2      # the reference's acceptance criterion (test/runtests.jl:31)
    prob.close()


def test_multi_theta_muse_matches_oracle(gpu, M, O):
    x, _ = O.sample_x_z("funnel", 2000, 9, M.DATA_SIM, [0.0] * 4)
    kw = dict(rng=1, nsims=20, maxsteps=4, get_covariance=True)
    a = M.muse(M.HipMuseProblem(x, model="funnel", ntheta=4, prior=M.GaussianPrior(0.0, 3.0)), [1.0] * 4, **kw)
    b = M.muse(OracleBatchedProblem(x, "funnel", 4, prior=M.GaussianPrior(0.0, 3.0)), [1.0] * 4, **kw)
    np.testing.assert_allclose(a.theta, b.theta, rtol=1e-8)
    np.testing.assert_allclose(a.J, b.J, rtol=1e-8)
    np.testing.assert_allclose(a.H, b.H, rtol=1e-6, atol=1e-6 * np.abs(b.H).max())
    np.testing.assert_allclose(a.Sigma, b.Sigma, rtol=1e-6, atol=1e-9)


def test_smooth_model_many_iterations_and_warm_history(gpu, M, O):
    N, th = 3000, [2.0, 3.0, 1.0, 2.5]
    prob = M.HipMuseProblem(None, model="smooth", ntheta=4, N=N)
    g, info = prob.map_and_score_batch(5, 0, 10, th, atol=1e-5, z0_mode=0)
    go, zo, io = O.map_and_score_batch("smooth", N, 5, 0, 10, th, atol=1e-5, z0_mode=0, nthreads=4)
    assert info["iterations"].min() > 10           # the L-BFGS ring buffer (m = 10) wraps
    assert np.array_equal(info["iterations"], io["iterations"]) and np.array_equal(info["f_calls"], io["f_calls"])
    assert np.array_equal(info["hist_words"], io["hist_words"])
    np.testing.assert_allclose(g, go, rtol=1e-10)
    np.testing.assert_allclose(prob.get_zhat(0, 10), zo, rtol=0, atol=1e-9)
    prob.close()


def test_full_size_config2_properties(gpu, M):
    """BASELINE.json configs[1] at full size (N = 10^4, nsims = 512): closed-form MAP and score for every sim,
    and the score moments E[s] = -N/(2(1+e^θ)), Var[s] = N e^{2θ}/(2(1+e^θ)²)."""
    N, S, theta = 10000, 512, 1.0
    prob = M.HipMuseProblem(None, model="funnel", ntheta=1, N=N)
    g, info = prob.map_and_score_batch(0, 0, S, [theta], atol=1e-2, z0_mode=0)
    assert np.all(info["status"] == 0) and np.all(info["iterations"] == 1) and np.all(info["f_calls"] == 3)
    e = np.exp(theta)
    mean, var = -N / (2 * (1 + e)), N * e**2 / (2 * (1 + e) ** 2)
    assert abs(g.mean() - mean) < 4 * np.sqrt(var / S)
    assert abs(g.var(ddof=1) / var - 1) < 4 * np.sqrt(2.0 / S)
    for k in (0, 17, 511):
        x, _ = prob.sample_x_z(M.SimRng(0, k), [theta])
        zh = prob.get_zhat(k, k + 1)[0]
        np.testing.assert_allclose(zh, x / (1 + np.exp(-theta)), rtol=0, atol=1e-10)
        np.testing.assert_allclose(g[k, 0], 0.5 * (np.exp(-theta) * np.sum(zh**2) - N), rtol=1e-12)
    # idempotence: a warm restart at the MAP needs no iteration and returns the same scores
    g2, info2 = prob.map_and_score_batch(0, 0, S, [theta], atol=1e-2, z0_mode=M.Z0_WARM)
    assert np.all(info2["iterations"] == 0) and np.all(info2["f_calls"] == 1)
    assert np.array_equal(g, g2)
    prob.close()


def test_full_size_config3_noise_1e6(gpu, M):
    """configs[2]: noise-level model, N = 10^6 (streaming policy): closed-form MAP and score."""
    N, theta = 1000000, 0.5
    prob = M.HipMuseProblem(None, model="noise", ntheta=1, N=N)
    g, info = prob.map_and_score_batch(3, 0, 4, [theta], atol=1e-2, z0_mode=0)
    assert np.all(info["status"] == 0)
    x, _ = prob.sample_x_z(M.SimRng(3, 2), [theta])
    zh = prob.get_zhat(2, 3)[0]
    np.testing.assert_allclose(zh, x / (1 + np.exp(theta)), rtol=0, atol=1e-9)
    np.testing.assert_allclose(g[2, 0], 0.5 * (np.exp(-theta) * np.sum((x - zh) ** 2) - N), rtol=1e-11)
    prob.close()


def test_results_do_not_depend_on_batch_geometry(gpu, M):
    """A sim's result depends on (seed, sim, theta) only: not on where it sits in a batch or how many
    workgroups share the launch (what makes a multi-GPU run equal the single-GPU run)."""
    prob = M.HipMuseProblem(None, model="funnel", ntheta=2, N=3001)
    th = [0.4, -0.3]
    g_all, _ = prob.map_and_score_batch(9, 0, 600, th)
    g_part, _ = prob.map_and_score_batch(9, 250, 260, th)
    assert np.array_equal(g_all[250:260], g_part)
    prob.close()


def test_c_abi_errors(gpu, M):
    with pytest.raises(ValueError):
        M.HipMuseProblem(None, model="nope", N=8)
    with pytest.raises(M.MuseError):
        M.HipMuseProblem(None, model="noise", ntheta=2, N=8)       # noise has one theta
    with pytest.raises(M.MuseError):
        M.HipMuseProblem(None, model="funnel", ntheta=65, N=80)    # > MUSE_MAX_THETA_EXT
    p = M.HipMuseProblem(None, model="funnel", N=64)
    with pytest.raises(M.MuseError) as e:
        p.map_and_score_batch(0, 0, 4, [0.0], include_data=True)   # no data set
    assert e.value.code == -3
    with pytest.raises(M.MuseError):
        p.map_and_score_batch_async(0, 5, 2, [0.0])                # bad range
    g, info = p.map_and_score_batch(0, 0, 0, [0.0])                # empty batch is fine
    assert g.shape == (0, 1)
    p.close()


def test_nonfinite_and_skip_errors(gpu, M):
    x = np.ones(64)
    x[3] = np.nan
    p = M.HipMuseProblem(x, model="funnel", N=None)
    with pytest.warns(RuntimeWarning):
        g, info = p.map_and_score_batch(0, 0, 2, [0.0], include_data=True)
        M.check_optim_soln(info)
    assert info["status"][0] == 5 and np.all(info["status"][1:] == 0)   # only the data element is poisoned
    p.close()


def test_rccl_single_rank(gpu, M):
    p = M.HipMuseProblem(None, model="funnel", N=64)
    uid = M.HipMuseProblem.comm_unique_id()
    p.comm_init(1, 0, uid)
    v = np.arange(12.0)
    assert np.array_equal(p.allgather_scores(v), v[None, :])
    assert np.array_equal(p.allreduce_sum(v), v)
    p.close()


def test_gathered_map_single_rank_matches_plain_map(gpu, M):
    """The sharded map body (solver -> device-side RCCL all-gather on a second stream -> pinned host), one
    rank: identical scores/infos to the plain batched map, padding rows zero, areas pipelined and reused."""
    N, nth, theta = 10000, 4, [1.0, 0.5, -0.5, 2.0]
    x = np.random.default_rng(3).normal(size=N)
    p = M.HipMuseProblem(x, model="funnel", ntheta=nth)
    p.comm_init(1, 0, M.HipMuseProblem.comm_unique_id())
    ref = [p.map_and_score_batch(0, b, b + 24, theta, include_data=(b == 0)) for b in (0, 24, 48, 72, 96, 120)]
    rows = 32
    pend = []
    out = []
    for k, b in enumerate((0, 24, 48, 72, 96, 120)):
        n = p.map_and_score_batch_gather_async(0, b, b + 24, theta, rows, include_data=(b == 0), result_area=k % 4)
        pend.append((n, k % 4))
        if len(pend) > 3:
            out.append(p.batch_wait_gathered(pend[0][0], rows, pend[0][1]) + (pend[0][0],))
            pend.pop(0)
    for n, area in pend:
        out.append(p.batch_wait_gathered(n, rows, area) + (n,))
    for (g_all, info, n), (g, inf) in zip(out, ref):
        assert g_all.shape == (1, rows, nth)
        assert np.array_equal(g_all[0, :n], g)
        assert np.all(g_all[0, n:] == 0.0)
        assert np.array_equal(info, inf)
    with pytest.raises(M.MuseError):
        p.batch_wait_gathered(1, rows, 0)          # nothing in flight on that area any more
    with pytest.raises(M.MuseError):
        p.map_and_score_batch_gather_async(0, 0, 24, theta, 8)   # rows_per_rank below the element count
    p.close()


def test_pipelined_launches_with_caller_owned_result_buffers(gpu, M):
    """The host loop of bench.py: batches enqueued on the four result areas ahead of their waits, every wait filling
    arrays the caller owns (`out=`).  The completion of an area is the solver dispatch's own signal; a launch that is
    still ahead in the stream must not make an earlier area look complete, and every step equals the blocking call."""
    N, nth, theta = 10000, 2, [0.7, -0.3]
    p = M.HipMuseProblem(None, model="funnel", ntheta=nth, N=N)
    ref = [p.map_and_score_batch(5, 100 * k, 100 * k + 300 + k, theta) for k in range(9)]
    bufs = {}
    pend, got = [], []
    for k in range(9):
        n = p.map_and_score_batch_async(5, 100 * k, 100 * k + 300 + k, theta, result_area=k % 4)
        pend.append((k, n))
        if len(pend) > 3:
            kk, nn = pend.pop(0)
            out = bufs.setdefault((kk % 4, nn), (np.empty((nn, nth)), np.zeros(nn, dtype=M._capi.INFO_DTYPE)))
            g, info = p.batch_wait(nn, kk % 4, out=out)
            assert g is out[0] and info is out[1]
            got.append((g.copy(), info.copy()))
    for kk, nn in pend:
        got.append(p.batch_wait(nn, kk % 4))
    for (g, info), (gr, ir) in zip(got, ref):
        assert np.array_equal(g, gr) and np.array_equal(info, ir)
    with pytest.raises(ValueError):
        p.batch_wait(3, 0, out=(np.empty((3, nth + 1)), np.zeros(3, dtype=M._capi.INFO_DTYPE)))
    p.close()


@pytest.mark.parametrize("model,N,nth,theta", [
    ("funnel", 512, 1, [0.3]), ("funnel", 10000, 4, [1.0, 0.5, -0.5, 2.0]), ("noise", 3001, 1, [0.4]),
    ("smooth", 2000, 3, [1.0, 2.0, 0.5]), ("funnel", 70001, 2, [0.3, 0.1]), ("smooth", 66001, 2, [1.0, 2.5])])
def test_implicit_diff_H(gpu, M, O, model, N, nth, theta):
    """Row f1: get_H! implicit-differentiation branch (src/muse.jl:335-405) against the oracle, and the
    oracle's own cross-check against the finite-difference Jacobian."""
    prob = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
    Hs, its = prob.implicit_H_batch(5, 0, 4, theta, atol=1e-1, cg_maxiter=100)
    for s in range(4):
        Ho, io = O.implicit_H(model, N, 5, s, theta, atol=1e-1, cg_maxiter=100)
        assert np.array_equal(its[s], io)
        np.testing.assert_allclose(Hs[s], Ho, rtol=1e-9, atol=1e-9 * np.abs(Ho).max())
    # tight MAP: implicit differentiation agrees with central differences of the same map
    Hs2, _ = prob.implicit_H_batch(5, 0, 1, theta, atol=1e-10)
    _, zfid, _ = O.map_and_score_batch(model, N, 5, 0, 1, theta, atol=1e-12, z0_mode=0)
    Hfd = O.fd_jacobian(model, N, 5, 0, theta, [1e-5] * nth, zfid[0], atol=1e-12)
    np.testing.assert_allclose(Hs2[0], Hfd, rtol=2e-6, atol=2e-6 * np.abs(Hfd).max())
    prob.close()


def test_get_H_implicit_diff_driver(gpu, M, O):
    x, _ = O.sample_x_z("funnel", 2000, 9, M.DATA_SIM, [0.0, 0.0])
    prob = M.HipMuseProblem(x, model="funnel", ntheta=2, prior=M.GaussianPrior(0.0, 3.0))
    res = M.muse(prob, [0.5, 0.5], rng=2, nsims=30, maxsteps=4)
    M.get_J_(res, prob, nsims=30)
    M.get_H_(res, prob, nsims=5, implicit_diff=True)
    a = res.H.copy()
    assert len(res.metadata["implicit_diff_cg_hists"]) == 5
    res.Hs, res.H = [], None
    M.get_H_(res, prob, nsims=5)          # finite-difference branch on the same streams
    np.testing.assert_allclose(a, res.H, rtol=0.05, atol=0.05 * np.abs(res.H).max())
    assert res.Sigma.shape == (2, 2)
    prob.close()


_SHARD_WORKER = r"""
import os, sys, pickle
sys.path.insert(0, {root!r})
import numpy as np, torch.distributed as dist
import museinference_jl_amd as M
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=world)
x = np.load({xfile!r})
local = M.HipMuseProblem(x, model="funnel", ntheta=2, prior=M.GaussianPrior(0.0, 3.0), device=0)
prob = M.ShardedMuseProblem(local)
assert prob.engine_comm and prob.transport == "shm" and local.comm_transport() == "shm"   # same host: shared memory
res = M.muse(prob, [1.0, 0.5], rng=3, nsims=21, maxsteps=4, get_covariance=True)
with open({out!r} + str(rank), "wb") as f:
    pickle.dump(dict(theta=res.theta, J=res.J, H=res.H, Sigma=res.Sigma, gs=np.array(res.gs)), f)
dist.destroy_process_group()
"""


def test_two_ranks_on_one_gpu_match_single_process(gpu, M, O, tmp_path):
    """The sharded path on real kernels: two processes (gloo group, both on GPU 0) each solve their block of
    the sims and exchange through the engine's shared-memory transport (the default for ranks of one host); every
    rank must reproduce the single-process result bit for bit."""
    import pickle
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    x, _ = O.sample_x_z("funnel", 3000, 5, M.DATA_SIM, [0.0, 0.0])
    xfile = str(tmp_path / "x.npy")
    np.save(xfile, x)
    out = str(tmp_path / "res")
    script = tmp_path / "worker.py"
    script.write_text(_SHARD_WORKER.format(root=root, port=29700 + os.getpid() % 200, xfile=xfile, out=out))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2"))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    single = M.HipMuseProblem(x, model="funnel", ntheta=2, prior=M.GaussianPrior(0.0, 3.0))
    ref = M.muse(single, [1.0, 0.5], rng=3, nsims=21, maxsteps=4, get_covariance=True)
    for r in range(2):
        got = pickle.load(open(out + str(r), "rb"))
        for k, want in (("theta", ref.theta), ("J", ref.J), ("H", ref.H), ("Sigma", ref.Sigma), ("gs", np.array(ref.gs))):
            assert np.array_equal(got[k], want), (r, k)
    single.close()


_NCCL_WORKER = r"""
import os, sys, pickle
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import museinference_jl_amd as M
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = np.load({xfile!r})
local = M.HipMuseProblem(x, model="funnel", ntheta=2, prior=M.GaussianPrior(0.0, 3.0), device=0)
prob = M.ShardedMuseProblem(local, transport="rccl")
assert prob.engine_comm and local.comm_transport() == "rccl"   # the exchange goes through the engine's RCCL communicator
res = M.muse(prob, [1.0, 0.5], rng=3, nsims=21, maxsteps=4, get_covariance=True)
with open({out!r}, "wb") as f:
    pickle.dump(dict(theta=res.theta, J=res.J, H=res.H, Sigma=res.Sigma, gs=np.array(res.gs)), f)
local.close()
dist.destroy_process_group()
"""


def test_sharded_driver_over_engine_rccl_one_rank(gpu, M, O, tmp_path):
    """ShardedMuseProblem on an nccl (RCCL) process group exchanges through muse_comm_* of the C ABI; with one
    rank the whole muse/get_J/get_H run must equal the unsharded run bit for bit."""
    import pickle
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    x, _ = O.sample_x_z("funnel", 3000, 5, M.DATA_SIM, [0.0, 0.0])
    xfile = str(tmp_path / "x.npy")
    np.save(xfile, x)
    out = str(tmp_path / "res")
    script = tmp_path / "worker.py"
    script.write_text(_NCCL_WORKER.format(root=root, port=29300 + os.getpid() % 200, xfile=xfile, out=out))
    p = subprocess.run([sys.executable, str(script)], timeout=300)
    assert p.returncode == 0
    single = M.HipMuseProblem(x, model="funnel", ntheta=2, prior=M.GaussianPrior(0.0, 3.0))
    ref = M.muse(single, [1.0, 0.5], rng=3, nsims=21, maxsteps=4, get_covariance=True)
    got = pickle.load(open(out, "rb"))
    for k, want in (("theta", ref.theta), ("J", ref.J), ("H", ref.H), ("Sigma", ref.Sigma), ("gs", np.array(ref.gs))):
        assert np.array_equal(got[k], want), k
    single.close()


@pytest.mark.parametrize("model,N,nth,theta0,prior", [
    ("funnel", 10000, 1, [1.0], "gauss"), ("funnel", 3000, 4, [1.0, 0.5, -0.5, 2.0], "gauss"),
    ("noise", 2000, 1, [0.8], "flat"), ("smooth", 1500, 2, [1.0, 0.3], "gauss"),
    ("funnel", 9999, 2, [0.8, 0.2], "gauss"), ("noise", 8191, 1, [0.3], "gauss")])   # odd N in the normals-cache layout
def test_native_outer_loop_equals_host_driver(gpu, M, O, model, N, nth, theta0, prior):
    """muse_run (the outer loop in the library's native host code) against the Python driver on the same
    launches: the same number of iterations, every history record and the final theta/J/H/Sigma to 1e-12
    (means and variances are summed in a different order: numpy's pairwise sums vs sequential)."""
    x, _ = O.sample_x_z(model, N, 11, M.DATA_SIM, [0.0] * nth)
    pr = M.GaussianPrior(0.0, 3.0) if prior == "gauss" else None
    res = {}
    for native in (False, True):
        prob = M.HipMuseProblem(x, model=model, ntheta=nth, prior=pr)
        res[native] = M.muse(prob, theta0, rng=5, nsims=64, maxsteps=12, theta_rtol=1e-3, get_covariance=True,
                             native=native)
        prob.close()
    a, b = res[False], res[True]
    assert len(a.history) == len(b.history) >= 3
    for ha, hb in zip(a.history, b.history):
        # (absolute tolerances relative to the scores' size: g_like' = g_dat' - mean(g_sims') is the difference of two numbers
        #  of that size, and the two loops sum the mean in different orders -- numpy pairwise, the library's 64-leaf tree)
        gscale = max(1.0, float(np.max(np.abs(np.asarray(ha["g_like_sims"])))))
        for k in ("θ", "θunreg", "θ′", "g_like_dat′", "g_like′", "g_prior′", "g_post′", "H⁻¹_post′", "H_prior′",
                  "H⁻¹_like′", "H⁻¹_like_sims′", "g_like_sims", "g_like_sims′"):
            np.testing.assert_allclose(np.asarray(hb[k]), np.asarray(ha[k]), rtol=1e-12, atol=1e-13 * gscale, err_msg=k)
        assert np.array_equal(hb["ẑ_history_sims"]["iterations"], ha["ẑ_history_sims"]["iterations"])
        assert hb["ẑ_history_dat"]["f_calls"] == ha["ẑ_history_dat"]["f_calls"]
    for k in ("theta", "J", "H", "Sigma"):
        # (relative to the matrix: the off-diagonal entries of a block model's H are differences of nearly equal numbers,
        #  10^-12 of the diagonal, and the two loops' thetas differ in the last bits)
        scale = float(np.max(np.abs(np.asarray(getattr(a, k)))))
        np.testing.assert_allclose(getattr(b, k), getattr(a, k), rtol=1e-10, atol=1e-10 * scale, err_msg=k)
    np.testing.assert_allclose(np.array(b.gs), np.array(a.gs), rtol=1e-12, atol=1e-13 * max(1.0, float(np.max(np.abs(np.array(a.gs))))))


def test_native_outer_loop_options(gpu, M, O):
    x, _ = O.sample_x_z("funnel", 2000, 3, M.DATA_SIM, [0.0])
    prob = M.HipMuseProblem(x, model="funnel", prior=M.GaussianPrior(0.0, 3.0))
    with pytest.raises(ValueError):
        M.muse(prob, [1.0], rng=1, nsims=16, alpha=lambda i: 0.5, native=True)   # not a plain option set
    r1 = M.muse(prob, [1.0], rng=1, nsims=16, maxsteps=3, z0=np.full(2000, 0.1), native=True)
    r2 = M.muse(prob, [1.0], rng=1, nsims=16, maxsteps=3, z0=np.full(2000, 0.1), native=False)
    np.testing.assert_allclose(r1.theta, r2.theta, rtol=1e-12)
    assert len(r1.history) == len(r2.history) >= 2
    prob.close()


# ---- several independent maps in one launch (muse_map_and_score_multi_async) ---------------------------------------
@pytest.mark.parametrize("model,N,nth,include_data", [("funnel", 10000, 1, False), ("funnel", 3000, 4, True),
                                                       ("noise", 700, 1, False), ("smooth", 900, 3, True)])
def test_multi_map_launch_equals_separate_maps(gpu, M, model, N, nth, include_data):
    """nmaps maps over the same elements, each with its own theta, in ONE launch (a rank's share of a strongly scaled map
    is smaller than the GPU: several maps resident at once fill it) give, bit for bit, what nmaps separate launches
    give: scores, solver infos and the resident MAPs (map m, element e at slot m*n + e)."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal(N)
    prob = M.HipMuseProblem(x, model=model, ntheta=nth)
    ref = M.HipMuseProblem(x, model=model, ntheta=nth)
    nmaps, nsims = 3, 11
    thetas = rng.uniform(-0.5, 1.0, size=(nmaps, nth))
    n = nsims + (1 if include_data else 0)
    tot = prob.map_and_score_multi_async(7, 2, 2 + nsims, thetas, include_data=include_data, atol=1e-3, result_area=1)
    assert tot == nmaps * n
    g, info = prob.batch_wait(tot, 1)
    Z = prob.get_zhat(0, tot)
    for m in range(nmaps):
        gm, im = ref.map_and_score_batch(7, 2, 2 + nsims, thetas[m], include_data=include_data, atol=1e-3)
        assert np.array_equal(g[m * n:(m + 1) * n], gm), m
        assert np.array_equal(info[m * n:(m + 1) * n], im), m
        assert np.array_equal(Z[m * n:(m + 1) * n], ref.get_zhat(0, n)), m
    # a plain launch afterwards is unaffected by the multi-map launch before it
    g1, i1 = prob.map_and_score_batch(7, 2, 2 + nsims, thetas[1], include_data=include_data, atol=1e-3)
    gm, im = ref.map_and_score_batch(7, 2, 2 + nsims, thetas[1], include_data=include_data, atol=1e-3)
    assert np.array_equal(g1, gm) and np.array_equal(i1, im)
    with pytest.raises(ValueError):
        prob.map_and_score_multi_async(7, 0, 4, np.zeros((9, nth)))     # more than MUSE_MAX_MAPS
    prob.close()
    ref.close()


def test_multi_map_launch_with_streaming_clusters(gpu, M):
    """Found by tools/fuzz_loops.py: streaming clusters (N > 10^4 with an element split, or N >= 65 536) draw the NEXT
    problem of a cluster during the current one's streaming passes -- with the theta that is in LDS; when a launch carries
    several maps the next problem may belong to another map and must then be drawn in the foreground."""
    for model, N, split, nmaps, nsims in [("noise", 19449, 4, 8, 34), ("funnel", 19730, 4, 8, 36), ("funnel", 70000, 0, 3, 40)]:
        prob = M.HipMuseProblem(None, model=model, ntheta=1, N=N)
        if split:
            prob.set_element_split(split)
        thetas = np.linspace(-0.8, 1.7, nmaps).reshape(nmaps, 1)
        tot = prob.map_and_score_multi_async(123, 3, 3 + nsims, thetas, atol=1e-2, result_area=1)
        g, info = prob.batch_wait(tot, 1)
        for m in range(nmaps):
            gm, im = prob.map_and_score_batch(123, 3, 3 + nsims, thetas[m], atol=1e-2)
            assert np.array_equal(g[m * nsims:(m + 1) * nsims], gm) and np.array_equal(info[m * nsims:(m + 1) * nsims], im), (model, N, m)
        prob.close()


def test_multi_map_with_element_split_and_gather_padding(gpu, M):
    """The same through the gathered entry (shared-memory transport, one rank) with padding rows between the maps'
    blocks, and with an element split (register-resident clusters) under the multi-map launch."""
    N, nth, nsims, nmaps, rows = 10000, 2, 5, 4, 8
    prob = M.HipMuseProblem(None, model="funnel", ntheta=nth, N=N)
    ref = M.HipMuseProblem(None, model="funnel", ntheta=nth, N=N)
    thetas = np.array([[0.1 * m, 0.3 - 0.2 * m] for m in range(nmaps)])
    prob.comm_init(1, 0, M.HipMuseProblem.comm_unique_id("shm"))
    for split in (0, 4):
        prob.set_element_split(split)
        ref.set_element_split(split)
        tot = prob.map_and_score_multi_gather_async(3, 0, nsims, thetas, rows, result_area=2)
        g_all, info = prob.batch_wait_gathered(tot, nmaps * rows, 2)
        assert g_all.shape == (1, nmaps * rows, nth)
        blk = g_all[0].reshape(nmaps, rows, nth)
        for m in range(nmaps):
            gm, im = ref.map_and_score_batch(3, 0, nsims, thetas[m])
            assert np.array_equal(blk[m, :nsims], gm), (split, m)
            assert np.all(blk[m, nsims:] == 0.0)
            assert np.array_equal(info[m * nsims:(m + 1) * nsims], im)
    prob.close()
    ref.close()


# ---- get_H! with any central_fdm(p, 1) and with FiniteDifferences' estimated step (muse_fd_values_columns) -----------
@pytest.mark.parametrize("model,N,nth", [("funnel", 3000, 3), ("funnel", 10000, 4), ("noise", 900, 1), ("smooth", 700, 2)])
def test_fd_values_columns_vs_oracle(gpu, M, O, model, N, nth):
    """The raw values of get_H!'s finite-difference map for arbitrary grids: offsets shared by the simulations and one row
    per (simulation, column) unit, a column range that begins and ends inside a simulation's Jacobian, an offset of zero,
    both fiducial modes -- against the oracle's per-simulation operators, and the central_fdm(3,1) entry rebuilt from them."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_problem import OracleBatchedProblem
    rng = np.random.default_rng(2)
    th0 = rng.uniform(-0.3, 0.8, size=nth)
    prob = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
    orc = OracleBatchedProblem(None, model, nth, N=N, nthreads=1)
    nsims, G, atol = 3, 5, 1e-6
    lo, hi = 1, nsims * nth - (1 if nth > 1 else 0)
    for fid_mode in (0, 1):
        shared = rng.uniform(-0.05, 0.05, size=(nth, G))
        shared[:, 2] = 0.0
        F, info = prob.fd_values_columns(9, 4, lo, hi, th0, shared, atol=atol, fid_mode=fid_mode)
        Fo, io = orc.fd_values_columns(9, 4, lo, hi, th0, shared, atol=atol, fid_mode=fid_mode)
        assert F.shape == (hi - lo, G, nth) and np.array_equal(info["status"], io["status"])
        np.testing.assert_allclose(F, Fo, rtol=1e-8, atol=1e-8 * np.abs(Fo).max())
        per = rng.uniform(-0.05, 0.05, size=(hi - lo, G))
        F2, _ = prob.fd_values_columns(9, 4, lo, hi, th0, per, per_unit=True, atol=atol, fid_mode=fid_mode)
        Fo2, _ = orc.fd_values_columns(9, 4, lo, hi, th0, per, per_unit=True, atol=atol, fid_mode=fid_mode)
        np.testing.assert_allclose(F2, Fo2, rtol=1e-8, atol=1e-8 * np.abs(Fo2).max())
        # the explicit-step central_fdm(3,1) entry is the same map
        step = np.full(nth, 0.03)
        cols, _ = prob.fd_jacobian_columns(9, 4, lo, hi, th0, step, atol=atol, fid_mode=fid_mode)
        Fpm, _ = prob.fd_values_columns(9, 4, lo, hi, th0, np.stack([step, -step], axis=1), atol=atol, fid_mode=fid_mode)
        want = (-0.5 * Fpm[:, 1] + 0.5 * Fpm[:, 0]) / step[(lo + np.arange(hi - lo)) % nth][:, None]
        assert np.array_equal(cols, want)
    with pytest.raises(ValueError):
        prob.fd_values_columns(9, 0, 0, nth, th0, np.zeros((nth + 1, 2)))
    prob.close()


def test_get_H_other_orders_and_estimated_step_on_hip(gpu, M, O):
    """get_H! on HipMuseProblem with fdm = central_fdm(5,1) (explicit step) and with neither step nor result.gs (the
    method estimates its step per simulation and column: src/muse.jl:300,411-413 on a fresh MuseResult) against the same
    driver on the oracle; and a sharded-style column range through the driver's batched path."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_problem import OracleBatchedProblem
    N, nth = 2000, 2
    x, _ = O.sample_x_z("funnel", N, 4, M.DATA_SIM, np.zeros(nth))
    th0 = np.array([0.5, -0.2])
    hip = M.HipMuseProblem(x, model="funnel", ntheta=nth, prior=M.GaussianPrior(0.0, 3.0))
    orc = OracleBatchedProblem(x, "funnel", nth, prior=M.GaussianPrior(0.0, 3.0), nthreads=1)
    for fdm, step, rtol in [("central_fdm(5,1)", [0.05, 0.03], 1e-7), ("central_fdm(2,1)", [0.02, 0.02], 1e-7),
                            ("central_fdm(3,1)", None, 2e-3), ("central_fdm(5,1)", None, 2e-3)]:
        rh, ro = M.MuseResult(theta=th0.copy()), M.MuseResult(theta=th0.copy())
        M.get_H_(rh, hip, rng=8, nsims=3, fdm=fdm, step=step, grad_z_logLike_atol=1e-10)
        M.get_H_(ro, orc, rng=8, nsims=3, fdm=fdm, step=step, grad_z_logLike_atol=1e-10)
        np.testing.assert_allclose(np.array(rh.Hs), np.array(ro.Hs), rtol=rtol, atol=rtol * np.abs(np.array(ro.Hs)).max())
    # closed form of the Gaussian model's H (estimated step, tight MAP): 1/2 e^-theta sigma^2 sum x z per block
    r = M.MuseResult(theta=th0.copy())
    M.get_H_(r, hip, rng=8, nsims=4, fdm="central_fdm(5,1)", grad_z_logLike_atol=1e-12)
    for s in range(4):
        xs, zs = hip.sample_x_z(M.SimRng(8, s), th0)
        sig = 1 / (1 + np.exp(-th0))
        want = [0.5 * np.exp(-th0[k]) * sig[k] ** 2 * np.sum((xs * zs)[N // 2 * k:N // 2 * (k + 1)]) for k in range(nth)]
        np.testing.assert_allclose(np.diag(r.Hs[s]), want, rtol=1e-4)
    hip.close()


@pytest.mark.parametrize("model,N,nth,split", [("funnel", 10000, 1, 0), ("funnel", 10000, 4, 4), ("noise", 70000, 1, 0),
                                                ("smooth", 3000, 2, 0), ("funnel", 400, 2, 0)])
def test_concurrent_lanes_give_the_same_results(gpu, M, model, N, nth, split):
    """muse_set_concurrency: result area r on lane r mod n (a stream, scratch, ticket counter and cluster state of its
    own), consecutive launches overlapping on the GPU -- the same bits as one launch after the other, for every
    placement kind (resident, split clusters, streaming clusters, streaming), different maps in flight at once."""
    prob = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
    ref = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
    if split:
        prob.set_element_split(split)
        ref.set_element_split(split)
    rng = np.random.default_rng(1)
    nel = 40 if N <= 10000 else 5
    thetas = [rng.uniform(-0.5, 1.0, size=nth) for _ in range(10)]
    want = [ref.map_and_score_batch(5, k, k + nel, th, atol=1e-3) for k, th in enumerate(thetas)]
    for lanes in (2, 4, 1):
        prob.set_concurrency(lanes)
        pend, got = [], {}
        for k, th in enumerate(thetas):
            n = prob.map_and_score_batch_async(5, k, k + nel, th, atol=1e-3, result_area=k % 4)
            pend.append((k, n))
            if len(pend) > 3:
                kk, nn = pend.pop(0)
                got[kk] = prob.batch_wait(nn, kk % 4)
        for kk, nn in pend:
            got[kk] = prob.batch_wait(nn, kk % 4)
        for k in range(len(thetas)):
            assert np.array_equal(got[k][0], want[k][0]) and np.array_equal(got[k][1], want[k][1]), (lanes, k)
    # a warm start and the native loop stay on lane 0 and see the MAPs of the last map
    prob.set_concurrency(2)
    g1, _ = prob.map_and_score_batch(5, 0, nel, thetas[0], atol=1e-3)
    g2, i2 = prob.map_and_score_batch(5, 0, nel, thetas[0], atol=1e-3, z0_mode=M.Z0_WARM)
    assert np.all(i2["iterations"] == 0) and np.array_equal(g1, g2)
    prob.close()
    ref.close()


def test_round3_entry_points_edge_cases(gpu, M, O):
    """Empty and degenerate inputs of the round-3 entry points: an empty sim range (with and without the data element), one
    map through the multi-map entry, N = 1 / ntheta = N, a finite-difference grid of one point at offset 0, bad arguments."""
    x = np.array([0.3, -1.2, 0.7, 0.1, 2.0])
    prob = M.HipMuseProblem(x, model="funnel", ntheta=5)               # ntheta = N: one element per block
    th = np.linspace(-0.5, 0.5, 5)
    assert prob.map_and_score_multi_async(1, 4, 4, np.stack([th, th + 0.1])) == 0          # nothing to do
    g, info = prob.batch_wait(0, 0)
    assert g.shape == (0, 5) and info.shape == (0,)
    tot = prob.map_and_score_multi_async(1, 4, 4, np.stack([th, th + 0.1]), include_data=True, result_area=3)
    g, info = prob.batch_wait(tot, 3)                                                        # two maps of the data element alone
    for m, t in enumerate((th, th + 0.1)):
        gm, im = prob.map_and_score_batch(1, 4, 4, t, include_data=True)
        assert np.array_equal(g[m:m + 1], gm) and np.array_equal(info[m:m + 1], im)
    tot = prob.map_and_score_multi_async(1, 0, 6, th[None, :])                              # one map through the multi entry
    g1, i1 = prob.batch_wait(tot, 0)
    g0, i0 = prob.map_and_score_batch(1, 0, 6, th)
    assert np.array_equal(g1, g0) and np.array_equal(i1, i0)
    prob.set_concurrency(3)
    assert prob.map_and_score_batch_async(1, 2, 2, th, result_area=2) == 0                   # empty batch on a lane
    prob.batch_wait(0, 2)
    with pytest.raises(M.MuseError):
        prob.set_concurrency(5)
    prob.set_concurrency(1)
    # finite differences: one grid point at offset 0 = the score at theta0 of the sim, MAP started from the fiducial MAP
    F, info = prob.fd_values_columns(1, 0, 0, 5, th, np.zeros((5, 1)), atol=1e-10)
    go, _, _ = O.map_and_score_batch("funnel", 5, 1, 0, 1, th, atol=1e-10, z0_mode=0)
    np.testing.assert_allclose(F[:, 0, :], np.tile(go[0], (5, 1)), rtol=1e-9, atol=1e-12)
    assert prob.fd_values_columns(1, 0, 3, 3, th, np.zeros((5, 2)))[0].shape == (0, 2, 5)    # empty column range
    with pytest.raises(M.MuseError):
        prob.fd_values_columns(1, 0, 0, 5, th, np.full((5, 1), np.nan))
    with pytest.raises(ValueError):
        prob.fd_values_columns(1, 0, 3, 2, th, np.zeros((5, 1)))
    prob.close()
    one = M.HipMuseProblem(np.array([0.4]), model="noise", ntheta=1)                        # N = 1
    def loop(dev):   # (a one-element problem is a wild iteration: whatever happens -- an answer or an error -- happens in both loops)
        try:
            n, theta, hist, gs, info = one.run_muse(3, [0.2], nsims=5, maxsteps=4, theta_rtol=0.0, atol=1e-6, alpha=0.5, device_loop=dev)
            return (n, theta.tobytes(), gs.tobytes())
        except M.MuseError as e:
            return str(e)
    assert loop(True) == loop(False)
    g, info = one.map_and_score_batch(3, 0, 5, [0.2], include_data=True, atol=1e-6)
    go, _, io = O.map_and_score_batch("noise", 1, 3, 0, 5, [0.2], atol=1e-6, x_data=np.array([0.4]), z0_mode=0)
    np.testing.assert_allclose(g, go, rtol=1e-10)
    assert np.array_equal(info["iterations"], io["iterations"])
    one.close()


def test_rccl_gather_with_lanes_enabled(gpu, M):
    """The device-side (RCCL) gather is tied to lane 0's stream order: with concurrency > 1, and after maps that ran on other
    lanes, a gathered map must still hand the collective the finished scores (one rank: the gathered block is the map's own)."""
    prob = M.HipMuseProblem(None, model="funnel", ntheta=2, N=10000)
    prob.set_concurrency(2)
    prob.comm_init(1, 0, M.HipMuseProblem.comm_unique_id("rccl"))
    th = np.array([0.4, -0.3])
    want, _ = M.HipMuseProblem(None, model="funnel", ntheta=2, N=10000).map_and_score_batch(9, 0, 300, th)
    for rep in range(6):
        for a in (1, 3):                                   # maps on lane 1 (areas 1 and 3) right before the gather
            prob.map_and_score_batch_async(9, 5, 505, th + 0.1 * rep, result_area=a)
        n = prob.map_and_score_batch_gather_async(9, 0, 300, th, 300, result_area=0)
        g_all, info = prob.batch_wait_gathered(n, 300, 0)
        assert np.array_equal(g_all[0], want), rep
        for a in (1, 3):
            prob.batch_wait(500, a)
    prob.close()
