"""Host drivers (muse_, get_J_, get_H_, finalize_result_) on CPU, driven through the oracle-backed
problem: reference semantics of src/muse.jl restated in SURVEY.md Appendix A."""
import math
import os

import numpy as np
import pytest

from oracle_problem import OracleBatchedProblem, OracleMuseProblem

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def funnel512(M, O):
    x, _ = O.sample_x_z("funnel", 512, 123, M.DATA_SIM, [0.0])
    return x


def test_muse_trajectory_matches_golden(M, O, funnel512):
    d = np.load(os.path.join(HERE, "golden", "muse_trajectory.npz"))
    assert np.array_equal(funnel512, d["x"])
    prob = OracleBatchedProblem(funnel512, "funnel", 1, prior=M.GaussianPrior(0.0, 3.0))
    res = M.muse(prob, [1.0], rng=42, nsims=32, get_covariance=True)
    np.testing.assert_allclose(np.array([h["θ"] for h in res.history]), d["thetas"], rtol=1e-13)
    np.testing.assert_allclose(res.theta, d["theta"], rtol=1e-12)
    np.testing.assert_allclose(res.J, d["J"], rtol=1e-12)
    np.testing.assert_allclose(res.H, d["H"], rtol=1e-10)
    np.testing.assert_allclose(res.Sigma, d["Sigma"], rtol=1e-10)
    np.testing.assert_allclose(np.array(res.Hs), d["Hs"], rtol=1e-10)
    np.testing.assert_allclose(np.array([h["g_like′"] for h in res.history]), d["g_like"], rtol=1e-12)
    np.testing.assert_allclose(np.array([h["H⁻¹_post′"] for h in res.history]), d["Hinv_post"], rtol=1e-12)


# keyword sets of tests/golden/make_golden.py::outer_loop_variants (the golden side is tests/muse_reference.py,
# an independent restatement of src/muse.jl:159-232; the product side is museinference.jl_amd/muse.py)
OUTER_VARIANTS = {
    "sims": dict(),
    "broyden": dict(Hinv_update="broyden"),
    "diagonal_broyden": dict(Hinv_update="diagonal_broyden"),
    "broyden_mem2": dict(Hinv_update="broyden", broyden_memory=2),
    "broyden_H0": dict(Hinv_update="broyden", Hinv_like0=np.diag([-0.02] * 4)),
    "alpha_regularize": dict(alpha=lambda i: 1.0 / (1 + i), regularize=lambda t: np.clip(t, -0.5, 0.8)),
}


class LogNormalVariancePrior:
    """log v ~ N(0, 3²) as a density over the variances v (analytic logpdf/grad/hess)."""

    def logpdf(self, v):
        return float(np.sum(-np.log(v) ** 2 / 18.0 - np.log(v)))

    def grad(self, v):
        return -np.log(v) / (9.0 * v) - 1.0 / v

    def hess(self, v):
        return np.diag((np.log(v) - 1.0) / (9.0 * v**2) + 1.0 / v**2)


def check_outer_variants(M, make_problem, rtol):
    """Rows f2/f4: every muse! keyword variant on the 4-block funnel against the independent restatement."""
    d = np.load(os.path.join(HERE, "golden", "muse_outer_variants.npz"))
    for name, kw in OUTER_VARIANTS.items():
        res = M.muse(make_problem(d["x"]), [1.0] * 4, rng=1, nsims=24, maxsteps=7, theta_rtol=0.0, native=False, **kw)
        thetas = np.array([h["θ"] for h in res.history])
        # result.θ is the UN-regularised iterate (src/muse.jl:230); the golden's last row is the same quantity
        np.testing.assert_allclose(thetas, d[name + "_thetas"][:-1], rtol=rtol, atol=rtol, err_msg=name)
        np.testing.assert_allclose(res.theta, d[name + "_thetas"][-1], rtol=rtol, atol=rtol, err_msg=name)
        for key, field in (("_Hinv_like", "H⁻¹_like′"), ("_Hinv_post", "H⁻¹_post′"), ("_g_post", "g_post′")):
            np.testing.assert_allclose(np.array([h[field] for h in res.history]), d[name + key], rtol=rtol, atol=rtol,
                                       err_msg=name + key)
    prob = M.PositiveThetaProblem(make_problem(d["x"], prior=None), prior=LogNormalVariancePrior())
    res = M.muse(prob, [np.e] * 4, rng=1, nsims=24, maxsteps=5, theta_rtol=0.0)
    np.testing.assert_allclose(np.array([h["θ"] for h in res.history]), d["positive_thetas"][:-1], rtol=rtol)
    np.testing.assert_allclose(res.theta, d["positive_thetas"][-1], rtol=rtol)
    np.testing.assert_allclose(np.array(res.gs), d["positive_gs"], rtol=rtol, atol=rtol)
    np.testing.assert_allclose(np.array([h["g_like′"] for h in res.history]), d["positive_g_like_t"], rtol=rtol, atol=rtol)


def test_outer_loop_variants_match_independent_restatement(M, O):
    def make(x, prior="gauss"):
        return OracleBatchedProblem(x, "funnel", 4, prior=M.GaussianPrior(0.0, 3.0) if prior == "gauss" else None)
    check_outer_variants(M, make, rtol=1e-10)


def test_convergence_test_domain_error(M, O, funnel512):
    """sqrt of a negative number in the convergence test is a DomainError in the reference (src/muse.jl:165): an
    H⁻¹_post′ that is not negative definite must not read as 'converged'."""
    prob = OracleBatchedProblem(funnel512, "funnel", 1, prior=M.GaussianPrior(0.0, 3.0))
    with pytest.raises(ValueError, match="DomainError"):
        M.muse(prob, [1.0], rng=0, nsims=8, maxsteps=5, Hinv_like0=[[+0.01]], Hinv_update="broyden", theta_rtol=1e-3)


def test_positive_theta_wrapper_forwards_only_safe_attributes(M, O, funnel512):
    """the variance front-end must not leak θ-space seams of the wrapped problem."""
    base = OracleBatchedProblem(funnel512, "funnel", 1)
    prob = M.PositiveThetaProblem(base, prior=LogNormalVariancePrior())
    assert prob.N == 512 and prob.ntheta == 1
    with pytest.raises(AttributeError):
        prob.run_muse
    v0 = np.array([1.3])
    Hs_v, its = prob.implicit_H_batch(3, 0, 2, v0, atol=1e-1, cg_maxiter=50)
    Hs_t, its_t = base.implicit_H_batch(3, 0, 2, np.log(v0), atol=1e-1, cg_maxiter=50)
    np.testing.assert_allclose(Hs_v, Hs_t / v0[0] ** 2, rtol=1e-14)
    # ... and that is the derivative the finite-difference branch takes in the variance space
    res = M.MuseResult(theta=v0.copy(), rng=3)
    M.get_J_(res, prob, nsims=12)
    M.get_H_(res, prob, nsims=2, implicit_diff=True)
    Himp = res.H.copy()
    res.Hs, res.H = [], None
    M.get_H_(res, prob, nsims=2, step=[1e-4], grad_z_logLike_atol=1e-10, fid_mode=1)
    res2 = M.MuseResult(theta=v0.copy(), rng=3, gs=list(res.gs))
    Hs_tight, _ = prob.implicit_H_batch(3, 0, 2, v0, atol=1e-10)
    np.testing.assert_allclose(res.H, Hs_tight.mean(axis=0), rtol=1e-5)
    assert abs(Himp[0, 0] / res.H[0, 0] - 1) < 0.05


def test_reference_acceptance_criterion(M, O, funnel512):
    """The reference's only numerical assertion (test/runtests.jl:31,56,81): |mu|/sigma < 2 at theta_true = 0,
    512-dim funnel, start theta0 = 1, prior N(0,3), nsims = 100, get_covariance."""
    prob = OracleBatchedProblem(funnel512, "funnel", 1, prior=M.GaussianPrior(0.0, 3.0))
    res = M.muse(prob, 1.0, rng=0, nsims=100, get_covariance=True)
    assert abs(res.dist.μ) / res.dist.σ < 2
    # closed form posterior width: (H J^-1 H + 1/9)^-1/2 with H = J = N/8 at theta = 0
    assert abs(res.dist.σ - 0.1249) < 0.02
    assert len(res.gs) == 100 and len(res.Hs) == 10  # get_H! runs nsims ÷ 10 sims (src/muse.jl:246)
    assert "MuseResult(" in repr(res) and "±" in repr(res)


def test_serial_path_equals_batched_path(M, O, funnel512):
    a = M.muse(OracleBatchedProblem(funnel512, prior=M.GaussianPrior()), [1.0], rng=7, nsims=8, maxsteps=3)
    b = M.muse(OracleMuseProblem(funnel512, prior=M.GaussianPrior(), batched=False), [1.0], rng=7, nsims=8, maxsteps=3)
    assert len(a.history) == len(b.history)
    for ha, hb in zip(a.history, b.history):
        np.testing.assert_allclose(ha["g_like′"], hb["g_like′"], rtol=1e-12)
    np.testing.assert_allclose(a.theta, b.theta, rtol=1e-12)


def test_get_J_reuses_gs_and_continues_streams(M, O, funnel512):
    prob = OracleBatchedProblem(funnel512, prior=M.GaussianPrior())
    res = M.muse(prob, [1.0], rng=5, nsims=16, maxsteps=2)
    gs0 = np.array(res.gs)
    M.get_J_(res, prob, nsims=16)                     # no new sims (src/muse.jl:499-502)
    assert np.array_equal(np.array(res.gs), gs0)
    np.testing.assert_allclose(res.J, np.atleast_2d(np.var(gs0, axis=0, ddof=1)), rtol=1e-13)
    M.get_J_(res, prob, nsims=24)                     # sims 16..23 are appended, evaluated at result.theta
    assert len(res.gs) == 24
    extra, _, _ = O.map_and_score_batch("funnel", 512, 5, 16, 24, res.theta, atol=1e-2, z0_mode=1)
    np.testing.assert_allclose(np.array(res.gs)[16:], extra, rtol=1e-13)


def test_get_H_step_default_and_mean(M, O, funnel512):
    prob = OracleBatchedProblem(funnel512, prior=M.GaussianPrior())
    res = M.MuseResult(theta=np.array([0.2]), rng=3)
    M.get_J_(res, prob, nsims=20)
    M.get_H_(res, prob, nsims=4)
    step = 0.1 / np.std(np.array(res.gs), axis=0, ddof=1)           # src/muse.jl:411-413
    _, zfid, _ = O.map_and_score_batch("funnel", 512, 3, M.MASTER_SIM, M.MASTER_SIM + 1, [0.2], atol=1e-2, z0_mode=0)
    want = [O.fd_jacobian("funnel", 512, 3, s, [0.2], step, zfid[0], atol=1e-2) for s in range(4)]
    np.testing.assert_allclose(np.array(res.Hs), np.array(want), rtol=1e-12)
    np.testing.assert_allclose(res.H, np.mean(want, axis=0), rtol=1e-12)
    H, J = res.H, res.J
    np.testing.assert_allclose(res.Sigma_inv, H.T @ np.linalg.inv(J) @ H + 1 / 9.0, rtol=1e-12)  # src/muse.jl:540
    assert res.time > 0


def test_multi_theta_and_update_modes(M, O):
    x, _ = O.sample_x_z("funnel", 400, 9, M.DATA_SIM, [0.0] * 4)
    prob = OracleBatchedProblem(x, "funnel", 4, prior=M.GaussianPrior(0.0, 3.0))
    base = M.muse(prob, [1.0] * 4, rng=1, nsims=24, maxsteps=6, get_covariance=True)
    assert base.Sigma.shape == (4, 4) and np.all(np.linalg.eigvalsh(base.Sigma) > 0)
    assert np.all(np.abs(base.theta) < 1.0)
    for mode in ("broyden", "diagonal_broyden"):
        r = M.muse(prob, [1.0] * 4, rng=1, nsims=24, maxsteps=6, Hinv_update=mode)
        assert np.all(np.isfinite(r.theta)) and np.all(np.abs(r.theta) < 1.5)
    r = M.muse(prob, [1.0] * 4, rng=1, nsims=24, maxsteps=4, alpha=lambda i: 0.5, regularize=lambda t: np.clip(t, -0.5, 2))
    assert np.all(r.history[-1]["θ"] >= -0.5)


def test_checkpoint_and_resume(M, O, funnel512, tmp_path):
    prob = OracleBatchedProblem(funnel512, prior=M.GaussianPrior())
    ck = str(tmp_path / "ck.pkl")
    kw = dict(rng=11, nsims=16, theta_rtol=0, grad_z_logLike_atol=1e-9)
    full = M.muse(OracleBatchedProblem(funnel512, prior=M.GaussianPrior()), [1.0], maxsteps=4, **kw)
    M.muse(prob, [1.0], maxsteps=2, checkpoint_filename=ck, **kw)
    res = M.load_result(ck)
    assert len(res.history) == 2 and res.rng == 11
    # resumes at len(history)+1 from result.theta/rng (src/muse.jl:134-135,159)
    M.muse_(res, prob, maxsteps=4, nsims=16, theta_rtol=0, grad_z_logLike_atol=1e-9)
    assert len(res.history) == 4
    # the resumed run restarts its MAPs from zero (src/muse.jl:151): same iterates up to the MAP tolerance
    np.testing.assert_allclose(res.theta, full.theta, atol=1e-6)


def test_save_maps_and_history_fields(M, O, funnel512):
    prob = OracleBatchedProblem(funnel512, prior=M.GaussianPrior())
    res = M.muse(prob, [1.0], rng=2, nsims=4, maxsteps=1, save_MAPs=True)
    h = res.history[0]
    for k in ("θ", "θunreg", "θ′", "θunreg′", "g_like_sims", "g_like_dat′", "g_like_sims′", "g_like′", "g_prior′",
              "g_post′", "H⁻¹_post′", "H_prior′", "H⁻¹_like′", "H⁻¹_like_sims′", "ẑ_history_dat", "ẑ_history_sims",
              "t", "ẑ_dat", "ẑ_sims"):
        assert k in h
    assert h["ẑ_dat"].shape == (512,) and len(h["ẑ_sims"]) == 4
    np.testing.assert_allclose(h["ẑ_dat"], funnel512 / (1 + np.exp(-1.0)), atol=1e-2)


def test_block_partition(M):
    for n in (0, 1, 7, 512, 513):
        for w in (1, 2, 3, 8):
            parts = [M.block_partition(10, 10 + n, w, r) for r in range(w)]
            assert parts[0][0] == 10 and parts[-1][1] == 10 + n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_priors(M):
    g = M.GaussianPrior(0.0, 3.0)
    c = M.CallablePrior(lambda t: -np.sum(t**2) / 18.0)
    th = np.array([0.7, -1.2])
    np.testing.assert_allclose(c.logpdf(th), g.logpdf(th), rtol=1e-12)
    np.testing.assert_allclose(c.grad(th), g.grad(th), rtol=1e-6)
    np.testing.assert_allclose(c.hess(th), g.hess(th), atol=1e-5)


def test_theta_transforms_and_self_consistency(M, O, funnel512):
    """Row f4: Transformedθ/UnTransformedθ (src/interface.jl:8-28) through a bounded-θ front-end whose
    parameters are the variances v = e^θ > 0, and check_self_consistency (src/interface.jl:209-230)."""
    base = OracleBatchedProblem(funnel512, "funnel", 1)
    prior_v = M.CallablePrior(lambda v: float(np.sum(-np.log(v) ** 2 / 18.0 - np.log(v))))  # N(0,3²) on log v
    prob = M.PositiveThetaProblem(base, prior=prior_v)
    res = M.check_self_consistency(prob, [1.7], atol=1e-2)
    assert max(res.values()) < 1e-4
    assert M.check_self_consistency(base, [0.3], atol=1e-2)["grad_chain_rule"] < 1e-8   # identity transform
    # muse in the transformed space lands where the untransformed-problem run does: v̂ = e^θ̂
    r_t = M.muse(prob, [np.e], rng=4, nsims=40, get_covariance=True)
    r_u = M.muse(OracleBatchedProblem(funnel512, "funnel", 1, prior=M.GaussianPrior(0.0, 3.0)), [1.0], rng=4, nsims=40)
    np.testing.assert_allclose(np.log(r_t.theta), r_u.theta, atol=2e-3)
    assert r_t.theta[0] > 0 and r_t.Sigma.shape == (1, 1)
    # chain rule on the Jacobian: H_v = H_θ′ / v² (scores and FD both in the untransformed space).  J is not
    # compared: result.gs are the scores at the θ *before* the last step (src/muse.jl:231,499-502), whose
    # 1/v factor differs from the final v̂.
    r_u = M.muse(OracleBatchedProblem(funnel512, "funnel", 1, prior=M.GaussianPrior(0.0, 3.0)), [1.0], rng=4, nsims=40,
                 get_covariance=True)
    np.testing.assert_allclose(r_t.H * r_t.theta[0] ** 2, r_u.H, rtol=1e-3)


def test_get_H_implicit_diff_host(M, O, funnel512):
    prob = OracleBatchedProblem(funnel512, prior=M.GaussianPrior())
    res = M.MuseResult(theta=np.array([0.2]), rng=3)
    M.get_J_(res, prob, nsims=20)
    M.get_H_(res, prob, nsims=6, implicit_diff=True, implicit_diff_cg_kwargs=dict(maxiter=50))
    want = np.mean([O.implicit_H("funnel", 512, 3, s, [0.2], atol=1e-1, cg_maxiter=50)[0] for s in range(6)], axis=0)
    np.testing.assert_allclose(res.H, want, rtol=1e-13)
    assert len(res.metadata["implicit_diff_cg_hists"]) == 6 and res.Sigma is not None
    M.get_H_(res, prob, nsims=6, implicit_diff=True)   # already have 6: no new sims (src/muse.jl:317-319)
    assert len(res.Hs) == 6


def test_bench_accounting_helpers():
    """bench.py's algorithmic-bytes accounting (words = 1 + 5E + Σ_k(4h_k + 4) + 2) and the
    thread budget of the CPU baseline."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(HERE), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import museinference_jl_amd as M
    from museinference_jl_amd import _capi
    info = np.zeros(3, dtype=_capi.INFO_DTYPE)
    info["f_calls"], info["iterations"], info["hist_words"] = [3, 3, 5], [1, 1, 2], [0, 0, 1]
    # funnel at theta != 0: E=3, K=1 -> 22 words; third sim E=5, K=2, one pair used -> 1 + 25 + (4 + 8) + 2 = 40
    assert bench.algorithmic_bytes(info, 10) == 8 * 10 * (22 + 22 + 40)
    # compulsory HBM bytes of the placement that ran: resident = zhat out (+ pairs and two-loop reads for K > 1);
    # streaming elementwise K = 1, E = 3 = 5 words (x, s | s, x, z: the second trial writes z + c s, the accepted step, into the MAP
    # slot and no last update pass runs -- round 5; 7 before: x, s | s, x | s, x, z)
    assert bench.compulsory_bytes(info, 10, "resident") == 8 * 10 * (1 + 1 + (1 + 2 + 4))
    assert bench.compulsory_bytes(info[:2], 10, "streaming") == 8 * 10 * (5 + 5)
    e2 = np.zeros(1, dtype=_capi.INFO_DTYPE)
    e2["f_calls"], e2["iterations"] = [2], [1]          # accepted at the first trial (fused into the sampler pass): the last pass runs
    assert bench.compulsory_bytes(e2, 10, "streaming") == 8 * 10 * (2 + 3)
    assert bench.compulsory_bytes(info[2:], 10, "stencil") == 8 * 10 * (6 + 4 + 3 * 4 + 11 + 4 + 8)
    assert bench.compulsory_bytes(info[2:], 10, "stencil_lds") == 8 * 10 * (6 + 3 + 2 * 4 + 9 + 3 + 4)
    assert len(bench.csrc_fingerprint()) == 16
    assert 1 <= bench.usable_cores(4) <= 4
    assert bench.measured_traffic("no_such_workload") is None
    assert set(bench.WORKLOADS) >= {"funnel_1e4", "funnel_512", "noise_1e6", "funnel4_1e4", "smooth_1e5", "cfg4_fd_H", "cfg5_smooth_1e5"}
    # the per-regime split of a loop's iteration times (round 5)
    hist = np.zeros((5, 9)); hist[:, -1] = [60e-6, 50e-6, 52e-6, 40e-6, 38e-6]
    inf = np.zeros((5, 4), dtype=_capi.INFO_DTYPE); inf["f_calls"] = [[3] * 4, [3] * 4, [3] * 4, [1] * 4, [1] * 4]
    reg = bench.iteration_regimes(hist, inf)
    assert reg["line_search"]["iterations"] == 2 and abs(reg["line_search"]["us_per_outer_iteration"] - 51.0) < 1e-9   # (the cold first one excluded)
    assert reg["converged_at_start"]["iterations"] == 2 and abs(reg["converged_at_start"]["us_per_outer_iteration"] - 39.0) < 1e-9


def test_native_path_selection(M, O, funnel512):
    """muse_(native="auto") only hands the loop to the library for problems that declare the capability; the
    oracle-backed problem (and any wrapper that merely forwards attributes) stays on the host driver."""
    from oracle_problem import OracleBatchedProblem
    prob = OracleBatchedProblem(funnel512, "funnel", 1, prior=M.GaussianPrior(0.0, 3.0))
    assert not getattr(type(prob), "supports_native_muse", False)
    r = M.muse(prob, [1.0], rng=0, nsims=8, maxsteps=2)          # "auto": falls through to the host driver
    assert len(r.history) == 2
    with pytest.raises(ValueError):
        M.muse(prob, [1.0], rng=0, nsims=8, maxsteps=2, native=True)
    # the sharded wrapper declares it too (muse_run_sharded), but only offers a native prior -- the switch muse_() looks at --
    # once the engine's own communicator is up; over torch collectives (here: no process group at all) it stays on the host driver
    assert M.HipMuseProblem.supports_native_muse and M.ShardedMuseProblem.supports_native_muse
    bare = M.ShardedMuseProblem.__new__(M.ShardedMuseProblem)
    bare.local, bare.engine_comm = prob, False
    assert bare.native_prior() is None


# ---- finite-difference methods of get_H! (fdm.py against the independent restatement in muse_reference.py) ----------
def test_fdm_coefficients_known_answers(M):
    from museinference_jl_amd.fdm import central_fdm
    import muse_reference as R
    known = {(2, 1): [-0.5, 0.5], (3, 1): [-0.5, 0.0, 0.5], (3, 2): [1.0, -2.0, 1.0],
             (5, 1): [1 / 12, -2 / 3, 0.0, 2 / 3, -1 / 12], (5, 3): [-0.5, 1.0, 0.0, -1.0, 0.5],
             (4, 1): [1 / 12, -2 / 3, 2 / 3, -1 / 12], (7, 1): [-1 / 60, 3 / 20, -3 / 4, 0.0, 3 / 4, -3 / 20, 1 / 60]}
    for (p, q), want in known.items():
        m = central_fdm(p, q)
        np.testing.assert_allclose(m.coefs, want, rtol=1e-15, atol=0)
        assert m.grid == R.fdm_central_grid(p) and list(m.coefs) == R.fdm_coefs(m.grid, q)
    m = central_fdm(3, 1)
    assert m.f_error_mult == 1.0 and abs(m.grad_magnitude_mult - 1 / 6) < 1e-16
    assert m.bound_estimator.grid == [-2, -1, 0, 1, 2] and m.bound_estimator.q == 3 and m.bound_estimator.bound_estimator is None
    with pytest.raises(ValueError):
        central_fdm(2, 2)
    # the exact-rational solve is done once per (grid, q) in a process and handed out read-only: it used to be most of the
    # wall time of a muse(get_covariance=True) at BASELINE's configs[1] (tools/profile_muse_py.py)
    assert central_fdm(3, 1).coefs is m.coefs and not m.coefs.flags.writeable
    with pytest.raises(ValueError):
        m.coefs[0] = 1.0


def test_fdm_estimated_step_and_derivatives_match_the_restatement(M):
    from museinference_jl_amd.fdm import central_fdm, as_fdm
    import muse_reference as R
    f = lambda t: [math.sin(3 * t) + 0.1 * t * t, math.exp(0.5 * t)]
    for p in (2, 3, 4, 5, 7):
        m = central_fdm(p, 1)
        fv = lambda offs: np.array([f(0.3 + o) for o in offs])
        h, hr = m.estimate_step(fv, 0.3), R.fdm_estimate_step(f, p, 1, 0.3)
        assert abs(h - hr) <= 1e-6 * hr, (p, h, hr)     # (the |f^(p)| estimate is a cancelling sum: its last digits depend on the summation order)
        assert 1e-10 < h < 1.0
        d, dr = m(f, 0.3), R.fdm_apply(f, p, 1, 0.3)
        np.testing.assert_allclose(d, dr, rtol=1e-9)
        np.testing.assert_allclose(d, [3 * math.cos(0.9) + 0.06, 0.5 * math.exp(0.15)], rtol=1e-5 if p <= 3 else 1e-8)
        np.testing.assert_allclose(m(f, 0.3, 1e-3), R.fdm_apply(f, p, 1, 0.3, 1e-3), rtol=1e-12)
    assert as_fdm("central_fdm(5, 1)").grid == [-2, -1, 0, 1, 2]
    # a constant function: the magnitudes vanish and the default step is taken
    assert central_fdm(3, 1).estimate_step(lambda offs: np.zeros((len(offs), 1))) == central_fdm(3, 1).default_step()


def test_get_H_other_orders_and_estimated_step(M, O, funnel512):
    """get_H! with fdm = central_fdm(5,1) and an explicit step, and with NEITHER a step nor result.gs (FiniteDifferences'
    own step estimation, src/muse.jl:300,411-413 -- a reachable reference path: get_H! on a fresh MuseResult): the batched
    seam (every grid point of every (sim, column) unit one problem of one launch) against the element-by-element path and
    against the independent restatement (muse_reference.fdm_apply on the oracle's per-simulation operators)."""
    import muse_reference as R
    from oracle_problem import OracleBatchedProblem, OracleMuseProblem
    x, _ = O.sample_x_z("funnel", 96, 9, M.DATA_SIM, [0.0, 0.0])
    th0, atol, nsims, seed = np.array([0.4, -0.3]), 1e-9, 2, 5
    mk = lambda cls, **kw: cls(x, "funnel", 2, prior=M.GaussianPrior(0.0, 3.0), **kw)

    def restated(p, step):
        zfid = O.map_and_score_batch("funnel", 96, seed, M.MASTER_SIM, M.MASTER_SIM + 1, th0, atol=atol, z0_mode=0)[1][0]
        Hs = []
        for s in range(nsims):
            cols = []
            for j in range(2):
                def f(eps):
                    t = th0.copy()
                    t[j] += eps
                    xs, _ = O.sample_x_z("funnel", 96, seed, s, t)
                    zh, _ = O.zhat_at_theta("funnel", xs, zfid, th0, atol)
                    return list(O.grad_theta("funnel", xs, zh, th0))
                cols.append(R.fdm_apply(f, p, 1, 0.0, None if step is None else step[j]))
            Hs.append(np.array(cols).T)
        return np.array(Hs)

    for p, step in [(5, np.array([0.05, 0.02])), (3, None), (5, None), (2, np.array([0.01, 0.01]))]:
        res = {}
        for name, prob in (("batched", mk(OracleBatchedProblem, nthreads=1)), ("serial", mk(OracleMuseProblem, batched=False))):
            r = M.MuseResult(theta=th0.copy())
            M.get_H_(r, prob, rng=seed, nsims=nsims, fdm=f"central_fdm({p},1)", step=step, grad_z_logLike_atol=atol)
            res[name] = np.array(r.Hs)
        want = restated(p, step)
        np.testing.assert_allclose(res["batched"], want, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(res["serial"], want, rtol=1e-9, atol=1e-9)
    # closed form: H_sim = 1/2 e^-theta sigma(theta)^2 sum x z per block (Gaussian model), the estimated step is accurate enough
    r = M.MuseResult(theta=th0.copy())
    M.get_H_(r, mk(OracleBatchedProblem, nthreads=1), rng=seed, nsims=1, fdm="central_fdm(5,1)", grad_z_logLike_atol=1e-12)
    xs, zs = O.sample_x_z("funnel", 96, seed, 0, th0)
    sig = 1 / (1 + np.exp(-th0))
    want = [0.5 * np.exp(-th0[k]) * sig[k] ** 2 * np.sum((xs * zs)[48 * k:48 * (k + 1)]) for k in range(2)]
    np.testing.assert_allclose(np.diag(r.Hs[0]), want, rtol=1e-5)
    with pytest.raises(ValueError):
        M.get_H_(M.MuseResult(theta=th0.copy()), mk(OracleBatchedProblem), rng=seed, nsims=1, fdm="central_fdm(3,2)")


def test_get_H_twice_adapted_method_falls_back_to_the_serial_path(M, O):
    """central_fdm(3, 1; adapt = 2) estimates the step of its bound estimator too: the batched seam is built for adapt = 1 and
    hands such a method to the element-by-element path (pjacobian, src/util.jl:9-27), which runs fdm(f, 0) as is."""
    from museinference_jl_amd.fdm import central_fdm
    from oracle_problem import OracleBatchedProblem
    x, _ = O.sample_x_z("funnel", 64, 2, M.DATA_SIM, [0.0])
    prob = OracleBatchedProblem(x, "funnel", 1, prior=M.GaussianPrior(0.0, 3.0), nthreads=1)
    m2 = central_fdm(3, 1, adapt=2)
    assert m2.bound_estimator.bound_estimator is not None
    r = M.MuseResult(theta=np.array([0.3]))
    M.get_H_(r, prob, rng=4, nsims=1, fdm=m2, grad_z_logLike_atol=1e-12)
    xs, zs = O.sample_x_z("funnel", 64, 4, 0, [0.3])
    sig = 1 / (1 + np.exp(-0.3))
    np.testing.assert_allclose(r.Hs[0][0, 0], 0.5 * np.exp(-0.3) * sig ** 2 * np.sum(xs * zs), rtol=1e-5)
    assert getattr(prob, "fd_maps_done", 0) == 0          # the batched seam was not used


def test_ignored_keywords_warn_once(M, O):
    from oracle_problem import OracleBatchedProblem
    import sys
    MU = sys.modules["museinference_jl_amd.muse"]     # (the package attribute `muse` is the function)
    MU._warned_ignored.clear()
    x, _ = O.sample_x_z("funnel", 32, 2, M.DATA_SIM, [0.0])
    prob = OracleBatchedProblem(x, "funnel", 1, prior=M.GaussianPrior(0.0, 3.0), nthreads=1)
    with pytest.warns(RuntimeWarning, match="pool"):
        M.muse(prob, [0.5], rng=1, nsims=4, maxsteps=1, pool="workers", native=False)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        M.muse(prob, [0.5], rng=1, nsims=4, maxsteps=1, pool="workers", native=False)    # said once, not again


def test_bench_algorithmic_valu_accounting():
    """bench.py's algorithmic VALU work of the resident placements (operation counts per element from the source x measured
    issue costs): the isotropic funnel's E = 3, K = 1 solve is 78 + 14 + 8 + 5 = 105 fp64 operations, 20 32x32->64 multiplies
    and 60 integer/select operations per element; a launch of 512 sims x 10^4 elements at 45.9 us and 2.4 GHz is ~0.5 of the
    issue peak -- and can never exceed 1."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(HERE), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    info = np.zeros(512, dtype=[("f_calls", "<i4"), ("iterations", "<i4"), ("hist_words", "<i4")])
    info["f_calls"], info["iterations"] = 3, 1
    cycles, fp64 = bench.algorithmic_valu(info, 10000, 2.4e9)
    assert fp64 == 512 * 10000 * 105
    per_elem = 105 * 4.4 + 20 * 4.75 + 60 * 2.7
    assert abs(cycles - 512 * 10000 * per_elem / 64) < 1e-6 * cycles
    frac = cycles / (1024 * 45.9e-6 * 2.4e9)
    assert 0.45 < frac < 0.56
    info["f_calls"], info["iterations"] = 9, 3          # two kept updates, six more trials
    _, fp64b = bench.algorithmic_valu(info, 10000, 2.4e9)
    assert fp64b == 512 * 10000 * (78 + 14 + 8 * 7 + 5 + 14 * 2)
