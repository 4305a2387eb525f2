"""The TWO-PARAMETER family of user-supplied models (include/muse_model.h, MUSE_MODEL_PAIR; round 6): blocks with a location AND a
scale parameter acting on the same elements -- what the reference's SimpleMuseProblem closures express freely (src/simple.jl:79-95)
and the one-parameter family (theta_k = the log-variance of one Gaussian factor) could not.  The shipped member,
museinference.jl_amd/models/normal_mean_var.h:

    z_i ~ N(mu_k, e^tau_k),  x_i ~ N(z_i, 1),   theta = (mu_0 .. mu_{K-1}, tau_0 .. tau_{K-1})

whose latent field integrates out in closed form, x_i ~ N(mu_k, 1 + e^tau_k): the MAP, the score and the whole muse() run have
known answers that owe nothing to this repository's arithmetic.  CPU: the checker's build of the header against the closed forms
and muse() on it against the exact posterior; GPU: the engine's library of the same header against the checker in every placement
(the draw bit for bit, equal iteration and evaluation counts, scores rtol 1e-10), the finite-difference get_H!, and muse() on HIP
against the exact posterior."""
import os

import numpy as np
import pytest
from scipy.optimize import root

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "museinference.jl_amd", "models", "normal_mean_var.h")
NAME = "normal_mean_var"
PRIOR_SIGMA = 3.0


def blocks(N, K):
    return (np.arange(N) * K) // N


def closed_forms(x, z, theta):
    """logLike, grad_z logLike, score of the model at (x, z, theta), written down."""
    K = len(theta) // 2
    k = blocks(x.size, K)
    mu, tau = np.asarray(theta)[k], np.asarray(theta)[K + k]
    iv = np.exp(-tau)
    f = -0.5 * (np.sum((x - z) ** 2 + iv * (z - mu) ** 2) + np.sum(tau))
    g = -(iv * (z - mu) - (x - z))
    s = np.array([np.sum((iv * (z - mu))[k == b]) for b in range(K)] +
                 [0.5 * (np.sum((iv * (z - mu) ** 2)[k == b]) - np.sum(k == b)) for b in range(K)])
    return f, g, s


def exact_posterior(x, K):
    """(mode, sigma) of theta = (mu_k, tau_k) given x_i ~ N(mu_k, 1 + e^tau_k) and the prior N(0, 3^2) on every component; sigma from
    the expected information at the mode (mu and tau are orthogonal: I_mumu = n / v, I_tautau = n/2 (e^tau / v)^2)."""
    k = blocks(x.size, K)
    mode, sigma = np.empty(2 * K), np.empty(2 * K)
    for b in range(K):
        xb = x[k == b]
        n = xb.size

        def grad(t):
            mu, tau = t
            v = 1.0 + np.exp(tau)
            d = xb - mu
            return [np.sum(d) / v - mu / PRIOR_SIGMA ** 2,
                    0.5 * np.exp(tau) / v ** 2 * (np.sum(d * d) - n * v) - tau / PRIOR_SIGMA ** 2]
        sol = root(grad, [xb.mean(), np.log(max(xb.var() - 1.0, 0.05))], tol=1e-13)
        assert sol.success
        mu, tau = sol.x
        v = 1.0 + np.exp(tau)
        mode[b], mode[K + b] = mu, tau
        sigma[b] = 1.0 / np.sqrt(n / v + 1.0 / PRIOR_SIGMA ** 2)
        sigma[K + b] = 1.0 / np.sqrt(0.5 * n * (np.exp(tau) / v) ** 2 + 1.0 / PRIOR_SIGMA ** 2)
    return mode, sigma


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("N,theta", [(1001, [0.7, -0.3, 0.4, 1.1]), (64, [-1.2, 0.3]), (4000, [0.1, 0.2, 0.3, 0.4, -0.5, 0.0, 0.5, 1.0])])
def test_checker_build_of_the_header_against_closed_forms(O, N, theta):
    theta = np.asarray(theta)
    K = theta.size // 2
    with O.user_model(HEADER, NAME):
        x, z = O.sample_x_z("user", N, 5, 3, theta)
        n1, n2 = O.normals(5, 3, N)
        k = blocks(N, K)
        np.testing.assert_allclose(z, theta[k] + np.exp(theta[K + k] / 2) * n1, rtol=1e-14, atol=1e-15)
        np.testing.assert_allclose(x, z + n2, rtol=0, atol=1e-15)
        zz = 0.7 * z + 0.1
        f, g = O.logLike_and_grad_z("user", x, zz, theta)
        fo, go, so = closed_forms(x, zz, theta)
        np.testing.assert_allclose(f, fo, rtol=1e-13)
        np.testing.assert_allclose(g, go, rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(O.grad_theta("user", x, zz, theta), so, rtol=1e-12, atol=1e-12)
        zh, info = O.zhat_at_theta("user", x, np.zeros(N), theta, atol=1e-9)
        iv = np.exp(-theta[K + k])
        np.testing.assert_allclose(zh, (x + iv * theta[k]) / (1 + iv), rtol=0, atol=1e-9)     # the MAP, in closed form
        # the finite-difference Jacobian of one simulation (src/muse.jl:426-442) against a numpy restatement with the same normals
        step = np.full(2 * K, 1e-3)
        H = O.fd_jacobian("user", N, 5, 3, theta, step, zh, atol=1e-10)

        def score_at(th_draw):
            zt = th_draw[k] + np.exp(th_draw[K + k] / 2) * n1
            xt = zt + n2
            zm = (xt + iv * theta[k]) / (1 + iv)
            return closed_forms(xt, zm, theta)[2]
        for j in range(2 * K):
            e = np.zeros(2 * K)
            e[j] = step[j]
            np.testing.assert_allclose(H[:, j], (score_at(theta + e) - score_at(theta - e)) / (2 * step[j]), rtol=1e-6, atol=1e-6)


def test_muse_on_the_checker_against_the_exact_posterior(M, O):
    """muse() end to end on the checker's build (no GPU): for this jointly Gaussian model MUSE is exact, so the estimate is the
    exact posterior mode up to the Monte-Carlo error of the simulation mean, and H^-1 J H^-T is the exact posterior variance."""
    from oracle_problem import OracleBatchedProblem
    N, K, truth, nsims = 3000, 1, np.array([0.8, 0.5]), 200
    with O.user_model(HEADER, NAME):
        x = O.sample_x_z("user", N, 99, (1 << 32) - 1, truth)[0]
        prob = OracleBatchedProblem(x, model="user", ntheta=2 * K, prior=M.GaussianPrior(0.0, PRIOR_SIGMA), nthreads=8)
        res = M.muse(prob, [0.0, 0.0], rng=20240, nsims=nsims, maxsteps=60, theta_rtol=1e-5, grad_z_logLike_atol=1e-7, alpha=1.0,
                     get_covariance=True)
    mode, sigma = exact_posterior(x, K)
    dev = np.abs(np.asarray(res.theta) - mode) / (sigma / np.sqrt(nsims))
    assert np.all(dev < 4.0), (res.theta, mode, dev)
    got = np.sqrt(np.diag(np.atleast_2d(res.Sigma)))
    assert np.all(np.abs(got / sigma - 1.0) < 5.0 * 0.5 * np.sqrt(2.0 / (nsims - 1)) + 0.03), (got, sigma)
    assert np.all(np.abs(mode - truth) / sigma < 4.0)


def test_check_model_consistency_of_the_pair_header_on_the_checker(M, O):
    """What AD guarantees in the reference (src/simple.jl:84-85) checked for the hand-written header: grad_z and BOTH kinds of score
    component (location, scale) against central differences of logLike, through the checker-backed problem."""
    from oracle_problem import OracleMuseProblem
    with O.user_model(HEADER, NAME):
        res = M.check_model_consistency(OracleMuseProblem(None, model="user", ntheta=4, N=2001), [0.4, -0.3, 0.9, 0.2], rng=5)
    assert res["grad_z"] <= 2e-5 + res["noise_floor"] and res["grad_theta"] <= 2e-5 + res["noise_floor"] and res["noise_floor"] < 1e-4


def test_header_kind_is_read_from_the_source(M):
    assert M.ElementwiseModel.packaged(NAME).pair is True
    assert M.ElementwiseModel.packaged("cubic").pair is False


# ------------------------------------------------------------------------------------------------ GPU
PLACEMENTS = [  # (N, ntheta = 2 K, theta, placement, split) -> every instantiation of the solver kernel the family gets
    (37, 2, [0.3, -0.2], -1, 0),                       # PlaceResident<256,1>, one block
    (300, 4, [0.4, -0.3, 0.9, 0.1], -1, 0),            # two blocks
    (2000, 2, [1.0, 0.5], -1, 0),                      # PlaceResident<512,4>
    (2000, 4, [0.5, -0.5, 0.2, 0.7], -1, 4),           # register clusters of 4 workgroups
    (10000, 2, [1.0, 1.0], -1, 0),                     # PlaceResident<512,10> (x, g in LDS): the headline's shape
    (10000, 4, [0.2, -0.3, -1.0, 0.0], -1, 0),
    (10000, 8, [0.2, -0.3, 1.0, 0.0, 0.5, -0.5, 0.0, 1.5], -1, 0),   # four blocks
    (10000, 2, [-0.7, 0.2], -1, 2),
    (9001, 6, [0.0, 0.7, -0.4, 0.3, 0.0, -0.3], -1, 8),
    (7001, 2, [-0.4, 0.6], 0, 0),                      # streaming, one workgroup
    (300, 2, [0.1, 0.2], 0, 0),                        # PlaceStreaming<256>
    (30011, 8, [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8], -1, 0),
    (70001, 4, [0.3, -0.2, 0.1, 0.9], -1, 0),          # N >= 65536: streaming clusters
]


def make(M, x, N, nth, placement, split, prior=None):
    prob = M.HipMuseProblem(x, model=M.ElementwiseModel.packaged(NAME), ntheta=nth, N=None if x is not None else N, prior=prior)
    if placement >= 0:
        prob.set_placement(placement)
    if split:
        prob.set_element_split(split)
    return prob


@pytest.mark.gpu
@pytest.mark.parametrize("N,nth,theta,placement,split", PLACEMENTS)
def test_pair_model_hip_against_the_checker(gpu, M, O, N, nth, theta, placement, split):
    from test_gpu_parity import assert_same_path_or_close
    theta = np.asarray(theta)
    with O.user_model(HEADER, NAME):
        truth = np.concatenate([np.full(nth // 2, 0.3), np.zeros(nth // 2)])
        xdata, _ = O.sample_x_z("user", N, 77, M.DATA_SIM, truth)
        prob = make(M, xdata, N, nth, placement, split)
        for sim in (0, 2**40 + 7):   # the draw, bit for bit
            x, z = prob.sample_x_z(M.SimRng(1234, sim), theta)
            xo, zo = O.sample_x_z("user", N, 1234, sim, theta)
            assert np.array_equal(z, zo) and np.array_equal(x, xo)
        zz = 0.7 * zo + 0.1
        f, g = prob.logLike_and_grad_z_logLike(xo, zz, theta)
        fo, go = O.logLike_and_grad_z("user", xo, zz, theta)
        np.testing.assert_allclose(f, fo, rtol=1e-12)
        np.testing.assert_allclose(g, go, rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(prob.grad_theta_logLike(xo, zz, theta), O.grad_theta("user", xo, zz, theta), rtol=1e-11, atol=1e-11)
        fc, gc, sc = closed_forms(xo, zz, theta)
        np.testing.assert_allclose(f, fc, rtol=1e-12)
        # the map of muse!: the data element and simulations, from zero, from the simulation's own z, warm at another theta
        nsims = 5 if N > 20000 else 13
        for z0_mode in (0, 1):
            g, info = prob.map_and_score_batch(42, 3, 3 + nsims, theta, include_data=True, atol=1e-6, z0_mode=z0_mode)
            go, zo, io = O.map_and_score_batch("user", N, 42, 3, 3 + nsims, theta, atol=1e-6, x_data=xdata, z0_mode=z0_mode)
            zh = prob.get_zhat(0, nsims + 1)
            same = assert_same_path_or_close(info, io, zh, zo, g, go, 1e-6, theta, "funnel", ctx=f"z0_mode {z0_mode}", g_rtol=1e-10)
            assert same.all(), (info["iterations"], io["iterations"], info["f_calls"], io["f_calls"])
            assert np.all(info["status"] == 0)
        # ... and the MAP in closed form: zhat = (x + iv mu) / (1 + iv)
        K = nth // 2
        k = blocks(N, K)
        iv = np.exp(-theta[K + k])
        np.testing.assert_allclose(zh[0], (xdata + iv * theta[k]) / (1 + iv), rtol=0, atol=2e-6)
        th2 = theta + 0.05
        g2, info2 = prob.map_and_score_batch(42, 3, 3 + nsims, th2, include_data=True, atol=1e-6, z0_mode=M.Z0_WARM)
        go2, zo2, io2 = O.map_and_score_batch("user", N, 42, 3, 3 + nsims, th2, atol=1e-6, x_data=xdata, z0_mode=2, zhat=zo.copy())
        assert np.array_equal(info2["f_calls"], io2["f_calls"])
        np.testing.assert_allclose(g2, go2, rtol=1e-10, atol=1e-9)
        # get_H! by finite differences (src/muse.jl:407-446): the draw at perturbed parameters carries the block's two sampling coefficients
        step = np.full(nth, 0.05)
        Hs, hinfo = prob.fd_jacobian_batch(11, 2, 5, theta, step, atol=1e-6)
        _, zfid, _ = O.map_and_score_batch("user", N, 11, M.MASTER_SIM, M.MASTER_SIM + 1, theta, atol=1e-6, z0_mode=0)
        for s in range(3):
            Ho = O.fd_jacobian("user", N, 11, 2 + s, theta, step, zfid[0], atol=1e-6)
            np.testing.assert_allclose(Hs[s], Ho, rtol=1e-7, atol=1e-7 * np.abs(Ho).max())
        prob.close()


@pytest.mark.gpu
def test_pair_model_refusals_and_routing(gpu, M, capfd):
    """An odd ntheta and more than MUSE_MAX_THETA parameters are refused at context creation; the implicit-differentiation get_H! is
    refused; the device-resident loop (one persistent launch for all iterations: the step's coefficient update on the device calls the
    HEADER's muse_model_coefs) gives the host loop's bits -- and is what ran."""
    model = M.ElementwiseModel.packaged(NAME)
    for nth in (1, 3, 10):
        with pytest.raises(M.MuseError, match="two parameters per block|MUSE_MAX_THETA"):
            M.HipMuseProblem(None, model=model, ntheta=nth, N=1000)
    for N, nth, th0 in ((5000, 2, [0.0, 0.0]), (10000, 4, [0.2, -0.1, 0.3, 0.0]), (3000, 8, [0.0] * 8), (400, 2, [0.1, 0.5])):
        x = np.sin(0.3 * np.arange(N)) + 0.4 + 0.8 * np.cos(1.7 * np.arange(N))
        prob = M.HipMuseProblem(x, model=model, ntheta=nth, prior=M.GaussianPrior(0.0, PRIOR_SIGMA))
        assert not prob.has_second_derivatives
        with pytest.raises(M.MuseError):
            prob.implicit_H_batch(1, 0, 2, th0)
        kw = dict(nsims=40, maxsteps=5, theta_rtol=0.0, atol=1e-6, alpha=0.7)
        b = prob.run_muse(3, th0, device_loop=False, **kw)
        capfd.readouterr()
        prob.debug_flags(M.HipMuseProblem.DEBUG_RUN_TIMING)
        a = prob.run_muse(3, th0, device_loop=True, **kw)
        prob.debug_flags(0)
        assert "[muse_run_device] launch call" in capfd.readouterr().err, (N, nth)      # the loop kernel, not muse_run's fall-back
        assert a[0] == b[0] == 5 and np.array_equal(a[1], b[1]) and np.array_equal(a[2][:, :-1], b[2][:, :-1]) and np.array_equal(a[3], b[3]), (N, nth)
        assert a[4].tobytes() == b[4].tobytes()
        c = prob.run_muse(3, a[1], device_loop=True, z0_warm=True, **dict(kw, maxsteps=3))        # continues from the resident MAPs
        d = prob.run_muse(3, a[1], device_loop=False, **dict(kw, maxsteps=5))                      # (leaves other MAPs behind)
        prob.run_muse(3, th0, device_loop=False, **kw)
        e2 = prob.run_muse(3, a[1], device_loop=False, z0_warm=True, **dict(kw, maxsteps=3))
        assert np.array_equal(c[1], e2[1]) and np.array_equal(c[3], e2[3]), (N, nth)
        if N != 400:
            prob.close()
    e = prob.model_eval(0.4, 0.6, 1.0, 0.7, 0.2, -0.3)      # (a, b) = (mu, tau): coefficients, draw, terms of the header on the host
    np.testing.assert_allclose([e["c0"], e["c1"], e["c2"], e["C"]], [0.4, np.exp(0.3), np.exp(-0.6), 0.6], rtol=1e-15)
    np.testing.assert_allclose([e["z"], e["x"], e["t0"], e["t1"]], [0.4 + np.exp(0.3) * 0.2, 0.4 + np.exp(0.3) * 0.2 - 0.3, 0.3, 0.09], rtol=1e-14)
    prob.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,truth,nsims", [(10000, 1, [0.8, 0.5], 512), (10000, 2, [0.8, -0.6, 0.5, 1.0], 256), (100000, 1, [-0.3, 0.2], 64)])
def test_muse_on_hip_against_the_exact_posterior_of_the_pair_model(gpu, M, N, K, truth, nsims):
    """muse() on the product path for the two-parameter model against the exact marginal posterior -- mode to the Monte-Carlo error
    of the simulation mean (4 sigma), reported covariance H^-1 J H^-T to the exact posterior variance -- through the native loop
    (muse_run) and through the Python loop: the same run."""
    model = M.ElementwiseModel.packaged(NAME)
    tmp = M.HipMuseProblem(None, model=model, ntheta=2 * K, N=N)
    x, _ = tmp.sample_x_z(M.SimRng(99, M.DATA_SIM), truth)
    tmp.close()
    mode, sigma = exact_posterior(x, K)
    prob = M.HipMuseProblem(x, model=model, ntheta=2 * K, prior=M.GaussianPrior(0.0, PRIOR_SIGMA))
    out = []
    for native in (True, False):
        res = M.muse(prob, [0.0] * (2 * K), rng=20240, nsims=nsims, maxsteps=60, theta_rtol=1e-5, grad_z_logLike_atol=1e-6, alpha=1.0,
                     get_covariance=True, native=native)
        dev = np.abs(np.asarray(res.theta) - mode) / (sigma / np.sqrt(nsims))
        assert np.all(dev < 4.0), (native, res.theta, mode, dev)
        got = np.sqrt(np.diag(np.atleast_2d(res.Sigma)))
        assert np.all(np.abs(got / sigma - 1.0) < 5.0 * 0.5 * np.sqrt(2.0 / (nsims - 1)) + 0.03), (native, got, sigma)
        out.append(np.asarray(res.theta))
    np.testing.assert_allclose(out[1], out[0], rtol=1e-9, atol=1e-12)
    assert np.all(np.abs(mode - np.asarray(truth)) / sigma < 4.5)
    prob.close()
