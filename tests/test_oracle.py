"""The CPU oracle against what pins it (no GPU): Philox known-answer vectors, closed forms of every
model, scipy's L-BFGS-B, finite differences, and the committed golden fixtures."""
import os

import numpy as np
import pytest
from scipy.optimize import minimize

HERE = os.path.dirname(os.path.abspath(__file__))


def test_philox_known_answers(O):
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        assert [int(v) for v in O.philox4x32_10(ctr, key)] == want


def test_normals_are_standard_normal_and_stream_only_depends_on_seed_sim(O):
    from scipy import stats
    n1, n2 = O.normals(3, 17, 200000)
    for n in (n1, n2):
        assert abs(n.mean()) < 0.01 and abs(n.std() - 1) < 0.01
        assert stats.kstest(n, "norm").pvalue > 1e-3
    assert abs(np.corrcoef(n1, n2)[0, 1]) < 0.01
    a, _ = O.normals(3, 17, 100)
    assert np.array_equal(a, n1[:100])          # prefix property: element i depends on (seed, sim, i) only
    b, _ = O.normals(3, 18, 100)
    assert not np.array_equal(a, b)


def test_normals_accuracy_against_numpy(O):
    # the fixed polynomial log / sincospi sequences are accurate to a few ulp
    N = 5000
    n1, n2 = O.normals(11, 0, N)
    w = np.array([O.philox4x32_10([i, 0, 0, 0], [11, 0]) for i in range(N)], dtype=np.uint64)
    k1 = (w[:, 0] << np.uint64(20)) | (w[:, 1] >> np.uint64(12))
    k2 = (w[:, 2] << np.uint64(20)) | (w[:, 3] >> np.uint64(12))
    u1 = (k1.astype(np.float64) + 0.5) * 2.0**-52
    u2 = (k2.astype(np.float64) + 0.5) * 2.0**-52
    r = np.sqrt(-2 * np.log(u1))
    np.testing.assert_allclose(n1, r * np.cos(2 * np.pi * u2), rtol=0, atol=5e-15)
    np.testing.assert_allclose(n2, r * np.sin(2 * np.pi * u2), rtol=0, atol=5e-15)


@pytest.mark.parametrize("theta", [-1.0, 0.0, 0.7, 2.0])
@pytest.mark.parametrize("N", [1, 7, 512])
def test_funnel_closed_forms(O, theta, N):
    x, z = O.sample_x_z("funnel", N, 0, 5, theta)
    n1, n2 = O.normals(0, 5, N)
    np.testing.assert_array_equal(z, O.exp_fixed(0.5 * theta) * n1)   # (the models' exp is a fixed sequence, within 1 ulp of libm)
    np.testing.assert_array_equal(x, z + n2)
    for z0 in (np.zeros(N), z):
        zh, info = O.zhat_at_theta("funnel", x, z0, theta, 1e-2)
        # isotropic Hessian (1+e^-θ) I: exact line minimisation along -g lands on the MAP in one iteration,
        # three evaluations (f(z0), the static guess alpha=1, the secant point)
        assert info["status"] == 0 and info["iterations"] <= 1 and info["f_calls"] <= 3
        np.testing.assert_allclose(zh, x / (1 + np.exp(-theta)), rtol=0, atol=1e-2 / (1 + np.exp(-theta)))
    zh, _ = O.zhat_at_theta("funnel", x, np.zeros(N), theta, 1e-10)
    zstar = x / (1 + np.exp(-theta))
    np.testing.assert_allclose(zh, zstar, rtol=0, atol=1e-12)
    np.testing.assert_allclose(O.grad_theta("funnel", x, zstar, theta),
                               0.5 * (np.exp(-theta) * np.sum(zstar**2) - N), rtol=1e-13)
    f, g = O.logLike_and_grad_z("funnel", x, z, theta)
    np.testing.assert_allclose(f, -0.5 * (np.sum((x - z) ** 2) + np.sum(z**2) / np.exp(theta) + N * theta), rtol=1e-13)
    np.testing.assert_allclose(g, (x - z) - np.exp(-theta) * z, rtol=1e-13, atol=1e-14)


@pytest.mark.parametrize("theta", [-0.5, 0.4])
def test_noise_closed_forms(O, theta):
    N = 300
    x, z = O.sample_x_z("noise", N, 1, 2, theta)
    zh, info = O.zhat_at_theta("noise", x, np.zeros(N), theta, 1e-10)
    np.testing.assert_allclose(zh, x / (1 + np.exp(theta)), rtol=0, atol=1e-12)
    s = O.grad_theta("noise", x, zh, theta)
    np.testing.assert_allclose(s, 0.5 * (np.exp(-theta) * np.sum((x - zh) ** 2) - N), rtol=1e-13)


def test_blocked_funnel_is_independent_funnels(O):
    N, th = 1000, [0.3, -0.7, 1.1, 0.0]
    x, z = O.sample_x_z("funnel", N, 4, 9, th)
    zh, _ = O.zhat_at_theta("funnel", x, np.zeros(N), th, 1e-10)
    blk = (np.arange(N) * 4) // N
    iv = np.exp(-np.asarray(th))[blk]
    np.testing.assert_allclose(zh, x / (1 + iv), rtol=0, atol=1e-11)
    s = O.grad_theta("funnel", x, zh, th)
    want = [0.5 * (np.exp(-th[k]) * np.sum(zh[blk == k] ** 2) - np.sum(blk == k)) for k in range(4)]
    np.testing.assert_allclose(s, want, rtol=1e-13)


def test_smooth_model_against_scipy_and_dense_solve(O):
    N, th = 400, [1.0, 2.0, 3.0, 0.5]
    x, z = O.sample_x_z("smooth", N, 0, 5, th)
    A = 0.5 * np.eye(N) + 0.25 * (np.roll(np.eye(N), 1, axis=1) + np.roll(np.eye(N), -1, axis=1))
    n1, n2 = O.normals(0, 5, N)
    np.testing.assert_allclose(x, A @ z + n2, rtol=0, atol=1e-14)
    blk = (np.arange(N) * 4) // N
    iv = np.exp(-np.asarray(th))[blk]
    zstar = np.linalg.solve(A.T @ A + np.diag(iv), A.T @ x)   # the MAP of the linear-Gaussian model
    zh, info = O.zhat_at_theta("smooth", x, np.zeros(N), th, 1e-9)
    assert info["iterations"] > 5  # non-isotropic: a real L-BFGS run
    # error <= ||g||_inf-ish / lambda_min(Hessian), lambda_min >= e^-3; the run may also stop on |df| = 0
    np.testing.assert_allclose(zh, zstar, rtol=0, atol=2e-6)
    fun = lambda v: tuple(-np.asarray(t) for t in O.logLike_and_grad_z("smooth", x, v, th))  # noqa: E731
    res = minimize(fun, np.zeros(N), jac=True, method="L-BFGS-B", options=dict(gtol=1e-10, ftol=1e-16, maxcor=10))
    np.testing.assert_allclose(res.x, zstar, rtol=0, atol=1e-5)


@pytest.mark.parametrize("model,th", [("funnel", [0.3, -0.2]), ("noise", [0.4]), ("smooth", [1.0, 2.0, 3.0, 0.5])])
def test_gradients_by_finite_differences(O, model, th):
    N = 200
    x, z = O.sample_x_z(model, N, 2, 3, th)
    rng = np.random.default_rng(0)
    zz, d = rng.normal(size=N), rng.normal(size=N)
    f0, g0 = O.logLike_and_grad_z(model, x, zz, th)
    h = 1e-6
    fd = (O.logLike_and_grad_z(model, x, zz + h * d, th)[0] - O.logLike_and_grad_z(model, x, zz - h * d, th)[0]) / (2 * h)
    np.testing.assert_allclose(g0 @ d, fd, rtol=1e-6)
    gt = O.grad_theta(model, x, zz, th)
    for j in range(len(th)):
        tp, tm = list(th), list(th)
        tp[j] += h
        tm[j] -= h
        fd = (O.logLike_and_grad_z(model, x, zz, tp)[0] - O.logLike_and_grad_z(model, x, zz, tm)[0]) / (2 * h)
        np.testing.assert_allclose(gt[j], fd, rtol=1e-5)


def test_score_moments_match_theory(O):
    # E[s] = -N/(2(1+e^θ)), Var[s] = N e^{2θ} / (2 (1+e^θ)^2)  
    N, theta, S = 512, 0.5, 400
    g, _, info = O.map_and_score_batch("funnel", N, 9, 0, S, [theta], atol=1e-2, nthreads=4)
    assert np.all(info["status"] == 0)
    e = np.exp(theta)
    mean, var = -N / (2 * (1 + e)), N * e**2 / (2 * (1 + e) ** 2)
    assert abs(g.mean() - mean) < 4 * np.sqrt(var / S)
    assert abs(g.var(ddof=1) / var - 1) < 4 * np.sqrt(2.0 / S)


def test_fd_jacobian_matches_closed_form(O):
    # per-sim common-random-number Jacobian of the funnel score: 1/2 e^{-θ0} σ(θ0)² d/dθ Σ x(θ)²
    N, th0, h = 512, 0.3, 1e-4
    x0, z0 = O.sample_x_z("funnel", N, 3, 1, [th0])
    _, zfid, _ = O.map_and_score_batch("funnel", N, 3, 1, 2, [th0], atol=1e-12, z0_mode=0)
    H = O.fd_jacobian("funnel", N, 3, 1, [th0], [h], zfid[0], atol=1e-12)
    sig = 1 / (1 + np.exp(-th0))
    want = 0.5 * np.exp(-th0) * sig**2 * np.sum(x0 * z0)
    np.testing.assert_allclose(H[0, 0], want, rtol=1e-6)


def test_oracle_reproduces_golden_fixtures(O):
    d = np.load(os.path.join(HERE, "golden", "per_sim.npz"))
    keys = sorted({k.rsplit("_", 1)[0] for k in d.files})
    assert len(keys) == 24
    for key in keys:
        model = "funnel" if key.startswith("funnel") else ("noise" if key.startswith("noise") else "smooth")
        N, seed = int(key.split("_N")[1].split("_")[0]), int(key.split("_s")[1])
        th = d[key + "_theta"]
        x, z = O.sample_x_z(model, N, seed, 5, th)
        assert np.array_equal(x, d[key + "_x"]) and np.array_equal(z, d[key + "_z"])
        zh, info = O.zhat_at_theta(model, x, np.zeros(N), th, 1e-2)
        assert np.array_equal(zh, d[key + "_zhat"])
        assert (info["iterations"], info["f_calls"]) == tuple(d[key + "_iters"][:2])
        assert np.array_equal(O.grad_theta(model, x, zh, th), d[key + "_score"])


def test_implicit_diff_H_closed_form_and_fd(O):
    """get_H! implicit-differentiation branch (src/muse.jl:335-405): funnel closed form
    H_sim = 1/2 e^-θ σ(θ)² Σ x z_true, and agreement with the finite-difference branch for every model."""
    N, th = 512, 0.3
    x, z = O.sample_x_z("funnel", N, 3, 1, [th])
    H, its = O.implicit_H("funnel", N, 3, 1, [th], atol=1e-12)
    sig = 1 / (1 + np.exp(-th))
    np.testing.assert_allclose(H[0, 0], 0.5 * np.exp(-th) * sig**2 * np.sum(x * z), rtol=1e-12)
    assert its[0] == 1  # isotropic Hessian: CG converges in one step
    for model, t in [("noise", [0.4]), ("funnel", [0.3, -0.2]), ("smooth", [1.0, 2.0, 0.5])]:
        _, zfid, _ = O.map_and_score_batch(model, 600, 5, 2, 3, t, atol=1e-12, z0_mode=0)
        # (step 1e-3: truncation error 1.6e-7 relative; at 1e-5 the stencil model's difference quotient amplifies the
        #  ~1e-9 noise an L-BFGS solve stops with -- x/f convergence before 1e-13 -- to a few 1e-6)
        Hfd = O.fd_jacobian(model, 600, 5, 2, t, [1e-3] * len(t), zfid[0], atol=1e-13)
        Him, its = O.implicit_H(model, 600, 5, 2, t, atol=1e-12)
        np.testing.assert_allclose(Him, Hfd, rtol=1e-6, atol=1e-6 * np.abs(Hfd).max())
    assert its.max() > 5   # the smooth model needs real CG iterations


def test_fixed_sequence_exp_is_within_one_ulp(O):
    """exp(theta/2), exp(-theta) are a fixed fdlibm-style sequence (so that the device-resident muse! loop and the host form
    the same bits): check it against libm over the whole range, the special values and the scaling paths."""
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-30, 30, 4000), rng.uniform(-745, 709, 2000), [0.0, -0.0, 1.0, -1.0, 0.5, 1e-300, -1e-300,
                         709.782712893384, 709.78271289338409, -708.3964185322641, -744.0, -745.13321910194, 1e-17, -1e-17,
                         0.34657359027997264, -0.34657359027997264]])
    for x in xs:
        got, want = O.exp_fixed(x), float(np.exp(x))
        if want == 0.0 or not np.isfinite(want):
            assert got == want or abs(got - want) <= 5e-324, x
        else:
            assert abs(got - want) <= 1.0 * np.spacing(want), (x, got, want)
    assert O.exp_fixed(710.0) == np.inf and O.exp_fixed(-746.0) == 0.0 and np.isnan(O.exp_fixed(np.nan))
    assert O.exp_fixed(0.0) == 1.0


def test_lbfgs_hagerzhang_reproduces_the_optim_documentation_example(O):
    """A known answer that does NOT come from this repository: the Optim.jl documentation ("Minimizing a multivariate
    function", docs/src/user/minimization.md) runs `optimize(f, g!, [0.0, 0.0], LBFGS())` on Rosenbrock's function
    f = (1 - x1)^2 + 100 (x2 - x1^2)^2 and prints `Iterations: 24`, `f(x) calls: 67`, `∇f(x) calls: 67` (default options:
    m = 10, InitialStatic(), HagerZhang(), g_tol = 1e-8).  The oracle's restatement of that solver -- the one every parity
    test compares the HIP path with -- reproduces both counters exactly.  (Quoted from memory: the documentation cannot
    be fetched in this image.  The counters are sensitive to every decision of the line search: one different
    bracketing step or Wolfe test moves them.)"""
    x, iterations, f_calls, status, f_min, gnorm = O.lbfgs_rosenbrock(1e-8)
    assert (iterations, f_calls) == (24, 67)
    assert status == 0 and gnorm <= 1e-8 and f_min < 1e-20
    np.testing.assert_allclose(x, [1.0, 1.0], rtol=0, atol=1e-9)
