"""A known answer for the WHOLE path that owes nothing to the oracle or to any arithmetic of this repository: for a
jointly Gaussian model MUSE is exact -- its score  g(x, zhat(x, theta)) - <g(x_sim, zhat(x_sim, theta))>  IS the gradient of the
marginal log-likelihood (the reference's premise, docs/src/index.md:13-18; Millea & Seljak 2022) -- and the three built-in
models are jointly Gaussian with marginals that can be written down:

    funnel, noise:   x_i ~ N(0, 1 + e^theta_k)                     L(theta) = -1/2 sum_i x_i^2 / (1 + e^theta_k) - 1/2 sum_k n_k log(1 + e^theta_k)
    smooth (1 theta): x ~ N(0, e^theta A A^T + I), A circulant      L(theta) = -1/2 sum_q |xhat_q|^2 / (N (1 + e^theta a_q^2)) - 1/2 sum_q log(1 + e^theta a_q^2)
                                                                    a_q = 1/2 + 1/2 cos(2 pi q / N)

So with S simulations muse() must return the root of  dL/dtheta + d logPrior/dtheta  up to the Monte-Carlo error of the
simulation mean, sigma_theta / sqrt(S) per component (sigma_theta^-2 = the expected information + the prior's), and the
covariance it reports, H^-1 J H^-T, must be sigma_theta^2 up to the error of a sample variance of S scores.  The reference's own
test only asks |theta - truth| / sigma < 2-3 (test/runtests.jl:31,56,81); this is S^(1/2) = 16-22 times sharper, and it pins
sampler, MAP, score, the muse! iteration, get_J! and get_H! at once.  CPU: the oracle-backed problem; GPU: HipMuseProblem at
BASELINE.json's configs[1] shape (N = 10^4, 512 sims)."""
import numpy as np
import pytest
from scipy.optimize import brentq

PRIOR_SIGMA = 3.0


def blocks(N, nth):
    return (np.arange(N) * nth) // N


def exact_scale_family(x, nth):
    """(posterior mode, sigma) per block of x_i ~ N(0, 1 + e^theta_k) with the prior theta_k ~ N(0, 3^2)."""
    k = blocks(x.size, nth)
    mode, sigma = np.empty(nth), np.empty(nth)
    for b in range(nth):
        xb = x[k == b]
        n, s2 = xb.size, float(np.sum(xb ** 2))
        f = lambda t: 0.5 * np.exp(t) / (1 + np.exp(t)) ** 2 * (s2 - n * (1 + np.exp(t))) - t / PRIOR_SIGMA ** 2
        mode[b] = brentq(f, -8.0, 8.0, xtol=1e-13)
        w = np.exp(mode[b]) / (1 + np.exp(mode[b]))
        sigma[b] = 1.0 / np.sqrt(0.5 * n * w ** 2 + 1.0 / PRIOR_SIGMA ** 2)
    return mode, sigma


def exact_smooth(x):
    N = x.size
    a2 = (0.5 + 0.5 * np.cos(2 * np.pi * np.arange(N) / N)) ** 2
    p = np.abs(np.fft.fft(x)) ** 2 / N          # |xhat_q|^2 / N: variance 1 + e^theta a_q^2 per mode
    f = lambda t: 0.5 * np.sum(np.exp(t) * a2 * (p - (1 + np.exp(t) * a2)) / (1 + np.exp(t) * a2) ** 2) - t / PRIOR_SIGMA ** 2
    mode = brentq(f, -8.0, 8.0, xtol=1e-13)
    w = np.exp(mode) * a2 / (1 + np.exp(mode) * a2)
    return np.array([mode]), np.array([1.0 / np.sqrt(0.5 * np.sum(w ** 2) + 1.0 / PRIOR_SIGMA ** 2)])


def check(M, prob, x, model, nth, nsims, theta0, atol, native="auto"):
    mode, sigma = exact_smooth(x) if model == "smooth" else exact_scale_family(x, nth)
    res = M.muse(prob, theta0, rng=20240, nsims=nsims, maxsteps=60, theta_rtol=1e-5, grad_z_logLike_atol=atol, alpha=1.0,
                 get_covariance=True, native=native)
    dev = np.abs(np.asarray(res.theta) - mode) / (sigma / np.sqrt(nsims))
    assert np.all(dev < 4.0), (res.theta, mode, dev)            # the MC error of the simulation mean, at 4 of its sigmas
    got = np.sqrt(np.diag(np.atleast_2d(res.Sigma)))
    # J is a sample covariance of nsims scores (relative error sqrt(2/(nsims-1))), H the mean of nsims/10 Jacobians whose
    # own scatter is O(sqrt(2/n_k)); sigma = sqrt(J)/H: five sigmas of the former
    assert np.all(np.abs(got / sigma - 1.0) < 5.0 * 0.5 * np.sqrt(2.0 / (nsims - 1)) + 0.02), (got, sigma)
    return res, mode, sigma


def oracle_data(O, model, N, nth, truth):
    return O.sample_x_z(model, N, 99, (1 << 32) - 1, truth)[0]


@pytest.mark.parametrize("model,N,nth,truth,nsims", [
    ("funnel", 2048, 1, [0.0], 256),            # the reference's documentation example (docs/src/index.md:56-69)
    ("funnel", 3000, 3, [1.0, -0.5, 0.3], 200),
    ("noise", 2048, 1, [0.7], 256),
    ("smooth", 600, 1, [1.5], 200),
])
def test_muse_on_the_oracle_against_the_exact_marginal_posterior(M, O, model, N, nth, truth, nsims):
    from oracle_problem import OracleBatchedProblem
    x = oracle_data(O, model, N, nth, truth)
    prob = OracleBatchedProblem(x, model=model, ntheta=nth, prior=M.GaussianPrior(0.0, PRIOR_SIGMA), nthreads=8)
    res, mode, sigma = check(M, prob, x, model, nth, nsims, [0.0] * nth, 1e-6 if model != "smooth" else 1e-7)
    assert np.all(np.abs(mode - np.asarray(truth)) / sigma < 4.0)   # (and the exact mode is where the data say it is)


@pytest.mark.gpu
@pytest.mark.parametrize("model,N,nth,truth,nsims", [
    ("funnel", 10000, 1, [1.0], 512),           # BASELINE.json configs[1]
    ("funnel", 10000, 4, [1.0, 0.0, -1.0, 2.0], 512),
    ("noise", 10000, 1, [0.5], 512),
    ("noise", 200000, 1, [-0.3], 128),          # streaming clusters
    ("smooth", 4096, 1, [1.0], 256),
    ("smooth", 100000, 1, [2.0], 64),           # the stencil model in clusters of 16
])
def test_muse_on_hip_against_the_exact_marginal_posterior(gpu, M, model, N, nth, truth, nsims):
    tmp = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
    x, _ = tmp.sample_x_z(M.SimRng(99, M.DATA_SIM), truth)
    tmp.close()
    prob = M.HipMuseProblem(x, model=model, ntheta=nth, prior=M.GaussianPrior(0.0, PRIOR_SIGMA))
    atol = 1e-6 if model != "smooth" else 1e-7
    res, mode, sigma = check(M, prob, x, model, nth, nsims, [0.0] * nth, atol, native=True)
    res2, _, _ = check(M, prob, x, model, nth, nsims, [0.0] * nth, atol, native=False)      # the Python loop: the same run
    np.testing.assert_allclose(res2.theta, res.theta, rtol=1e-9, atol=1e-12)
    prob.close()


def ensemble(M, make_prob, x, nth, nsims, nruns, atol=1e-6):
    """nruns muse() runs on the same data with independent master seeds: the standardized deviations from the exact mode
    (units of sigma/sqrt(nsims)), and J, H relative to the exact information (flat part: without the prior's 1/9)."""
    mode, sigma = exact_scale_family(x, nth)
    k = blocks(x.size, nth)
    n_k = np.array([(k == b).sum() for b in range(nth)])
    w = np.exp(mode) / (1 + np.exp(mode))
    F = 0.5 * n_k * w ** 2
    dev, J, H = [], [], []
    for run in range(nruns):
        prob = make_prob()
        res = M.muse(prob, [0.0] * nth, rng=5000 + run, nsims=nsims, maxsteps=60, theta_rtol=1e-5, grad_z_logLike_atol=atol, alpha=1.0,
                     get_covariance=True)
        dev.append((np.asarray(res.theta) - mode) / (sigma / np.sqrt(nsims)))
        J.append(np.diag(np.atleast_2d(res.J)) / F)
        H.append(np.diag(np.atleast_2d(res.H)) / F)
        if hasattr(prob, "close"):
            prob.close()
    return np.array(dev), np.array(J), np.array(H)


def assert_unbiased(dev, J, H, nsims):
    n = dev.size
    assert abs(dev.mean()) < 4.0 / np.sqrt(n), dev.mean()                 # the estimator scatters AROUND the exact mode ...
    assert 0.6 < dev.std() < 1.4, dev.std()                                # ... by sigma / sqrt(nsims), as the theory says
    assert abs(J.mean() - 1.0) < 4.0 * np.sqrt(2.0 / (nsims - 1)) / np.sqrt(J.size), J.mean()   # get_J!: the exact information
    assert abs(H.mean() - 1.0) < 0.01, H.mean()                            # get_H!: the same number, to 1 %


def test_muse_is_unbiased_around_the_exact_mode_on_the_oracle(M, O):
    """16 independent master seeds on the documentation example (2048-dim funnel): deviations from the exact posterior mode
    have mean 0 and standard deviation sigma/sqrt(nsims); get_J! and get_H! both return the exact Fisher information."""
    from oracle_problem import OracleBatchedProblem
    x = oracle_data(O, "funnel", 2048, 1, [0.0])
    dev, J, H = ensemble(M, lambda: OracleBatchedProblem(x, model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, PRIOR_SIGMA), nthreads=8),
                         x, 1, 256, 16)
    assert_unbiased(dev, J, H, 256)


@pytest.mark.gpu
def test_muse_is_unbiased_around_the_exact_mode_on_hip(gpu, M):
    """The same ensemble on the product path at configs[1]'s shape and with 4 blocks: 24 seeds x 512 sims x (1 | 4) theta."""
    for nth, truth in ((1, [1.0]), (4, [1.0, 0.0, -1.0, 2.0])):
        tmp = M.HipMuseProblem(None, model="funnel", ntheta=nth, N=10000)
        x, _ = tmp.sample_x_z(M.SimRng(99, M.DATA_SIM), truth)
        tmp.close()
        dev, J, H = ensemble(M, lambda: M.HipMuseProblem(x, model="funnel", ntheta=nth, prior=M.GaussianPrior(0.0, PRIOR_SIGMA)), x, nth, 512, 24)
        assert_unbiased(dev, J, H, 512)


@pytest.mark.gpu
@pytest.mark.parametrize("model,nth,truth", [("funnel", 1, [1.0]), ("funnel", 4, [1.0, 0.0, -1.0, 2.0]), ("noise", 1, [0.5])])
def test_both_get_H_branches_return_the_exact_information_on_hip(gpu, M, model, nth, truth):
    """get_H! by finite differences (src/muse.jl:407-446) and by implicit differentiation (src/muse.jl:335-405) at the exact
    posterior mode: both are the expected information N/2 (e^theta / (1 + e^theta))^2 per block, to 1 %."""
    N = 10000
    tmp = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
    x, _ = tmp.sample_x_z(M.SimRng(99, M.DATA_SIM), truth)
    tmp.close()
    mode, sigma = exact_scale_family(x, nth)
    w = np.exp(mode) / (1 + np.exp(mode))
    F = 0.5 * (N / nth) * w ** 2
    prob = M.HipMuseProblem(x, model=model, ntheta=nth, prior=M.GaussianPrior(0.0, PRIOR_SIGMA))
    for kw in (dict(step=0.1 * sigma), dict(implicit_diff=True)):
        res = M.MuseResult()
        res.theta = mode.copy()
        M.get_H_(res, prob, mode, rng=7, nsims=64, grad_z_logLike_atol=1e-6, **kw)
        H = np.atleast_2d(res.H)
        np.testing.assert_allclose(np.diag(H), F, rtol=1e-2, err_msg=str(kw))
        off = H - np.diag(np.diag(H))
        assert np.abs(off).max() <= 2e-2 * F.min(), kw          # blocks are independent: no cross terms beyond sampling noise
    prob.close()
