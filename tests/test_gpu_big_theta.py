"""ntheta > MUSE_MAX_THETA (the big tier of include/muse_hip.h: up to MUSE_MAX_THETA_EXT = 64 components; the reference has
no bound, src/muse.jl:296-333 maps over however many columns theta has).  The per-block coefficients come from the kernel-argument
segment, an element's block from arithmetic, the block sums eight at a time -- streaming placements only; the native muse!
loops and several maps per launch refuse it, muse() then runs its loop over the batched maps.

The HIP path against the CPU oracle exactly as the small tiers are checked: sampler bit for bit, identical iteration and
evaluation counts, MAPs and scores to rounding -- single workgroups and clusters, the elementwise and the stencil model, a
user-supplied model; the per-simulation operators; both get_H! branches; a whole muse() run."""
import os

import numpy as np
import pytest

from test_gpu_parity import assert_same_path_or_close

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CUBIC = os.path.join(ROOT, "museinference.jl_amd", "models", "cubic.h")


def thetas(nth, seed=0):
    return np.round(np.random.RandomState(seed).uniform(-1.0, 1.5, nth), 3)


CASES = [  # (model, N, ntheta, nsims, split)
    ("funnel", 10000, 9, 6, 0), ("funnel", 10000, 64, 6, 0), ("funnel", 9999, 33, 4, 0), ("funnel", 300, 12, 5, 0),
    ("funnel", 70001, 20, 3, 0), ("funnel", 10000, 16, 4, 4), ("smooth", 3001, 17, 4, 0), ("smooth", 66001, 10, 2, 0),
    ("funnel", 64, 64, 3, 0),
]


@pytest.mark.gpu
@pytest.mark.parametrize("model,N,nth,nsims,split", CASES)
def test_big_theta_map_against_oracle(gpu, M, O, model, N, nth, nsims, split):
    theta = thetas(nth, N + nth)
    prob = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
    if split:
        prob.set_element_split(split)
    # the draw, bit for bit (blocks by arithmetic = the oracle's floor(i ntheta / N))
    x, z = prob.sample_x_z(M.SimRng(11, 2), theta)
    xo, zo = O.sample_x_z(model, N, 11, 2, theta)
    assert np.array_equal(x, xo) and np.array_equal(z, zo)
    # the per-simulation operators
    zz = 0.7 * z + 0.01
    f, gz = prob.logLike_and_grad_z_logLike(x, zz, theta)
    fo, gzo = O.logLike_and_grad_z(model, x, zz, theta)
    np.testing.assert_allclose(f, fo, rtol=1e-12)
    np.testing.assert_allclose(gz, gzo, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(prob.grad_theta_logLike(x, zz, theta), O.grad_theta(model, x, zz, theta), rtol=1e-11, atol=1e-11)
    # the batched map
    g, info = prob.map_and_score_batch(11, 0, nsims, theta, atol=1e-6, z0_mode=0)
    go, zho, io = O.map_and_score_batch(model, N, 11, 0, nsims, theta, atol=1e-6, z0_mode=0)
    zh = prob.get_zhat(0, nsims)
    same = assert_same_path_or_close(info, io, zh, zho, g, go, 1e-6, theta, model, ctx=f"{model} N={N} ntheta={nth}")
    assert same.all()
    assert g.shape == (nsims, nth) and np.all(info["status"] == 0)
    prob.close()


@pytest.mark.gpu
@pytest.mark.parametrize("model,N,nth", [("funnel", 6000, 12), ("smooth", 2500, 9), ("funnel", 66000, 10)])
def test_big_theta_get_H_branches_against_oracle(gpu, M, O, model, N, nth):
    theta = thetas(nth, 5)
    step = np.full(nth, 0.05)
    prob = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
    Hs, info = prob.fd_jacobian_batch(9, 0, 2, theta, step, atol=1e-4)
    _, zfid, _ = O.map_and_score_batch(model, N, 9, M.MASTER_SIM, M.MASTER_SIM + 1, theta, atol=1e-4, z0_mode=0)
    for s in range(2):
        Ho = O.fd_jacobian(model, N, 9, s, theta, step, zfid[0], atol=1e-4)
        np.testing.assert_allclose(Hs[s], Ho, rtol=1e-8, atol=1e-8 * np.abs(Ho).max())
    Hi, its = prob.implicit_H_batch(5, 0, 2, theta, atol=1e-1, cg_maxiter=100)
    for s in range(2):
        Ho, io = O.implicit_H(model, N, 5, s, theta, atol=1e-1, cg_maxiter=100)
        assert np.array_equal(its[s], io)
        np.testing.assert_allclose(Hi[s], Ho, rtol=1e-9, atol=1e-9 * np.abs(Ho).max())
    cols, ci = prob.implicit_H_columns(5, 0, 3, nth + 4, theta)
    np.testing.assert_allclose(cols, np.concatenate([Hi[0].T, Hi[1].T])[3:nth + 4], rtol=1e-12, atol=1e-12 * np.abs(Hi).max())
    prob.close()


@pytest.mark.gpu
def test_big_theta_user_model(gpu, M, O):
    N, nth = 8000, 11
    theta = thetas(nth, 3) * 0.5
    with O.user_model(CUBIC, "cubic"):
        prob = M.HipMuseProblem(None, model=M.ElementwiseModel.packaged("cubic"), ntheta=nth, N=N)
        g, info = prob.map_and_score_batch(4, 0, 4, theta, atol=1e-6, z0_mode=0)
        go, zho, io = O.map_and_score_batch("user", N, 4, 0, 4, theta, atol=1e-6, z0_mode=0)
        zh = prob.get_zhat(0, 4)
        assert_same_path_or_close(info, io, zh, zho, g, go, 1e-6, theta, "user", ctx="cubic, 11 components", z_atol=1e-8, g_rtol=1e-9)
        Hi, its = prob.implicit_H_batch(5, 0, 2, theta)
        for s in range(2):
            Ho, io = O.implicit_H("user", N, 5, s, theta, atol=1e-1, cg_maxiter=100)
            assert np.all(np.abs(its[s] - io) <= 1)
            np.testing.assert_allclose(Hi[s], Ho, rtol=1e-7, atol=1e-7 * np.abs(Ho).max())
        prob.close()


@pytest.mark.gpu
def test_big_theta_whole_run_and_refusals(gpu, M, O):
    """muse() + get_J! + get_H! at 12 components: the native loops refuse (MUSE_MAX_THETA), muse() runs its loop over the batched
    maps -- against the same driver over the CPU oracle -- and the estimate sits where the exact posterior of the funnel says."""
    from oracle_problem import OracleMuseProblem
    N, nth = 6000, 12
    truth = thetas(nth, 8) * 0.6
    x, _ = O.sample_x_z("funnel", N, 21, M.DATA_SIM, truth)
    prob = M.HipMuseProblem(x, model="funnel", ntheta=nth, prior=M.GaussianPrior(0.0, 3.0))
    assert prob.native_prior() is None
    with pytest.raises(M.MuseError) as e:
        prob.run_muse(3, np.zeros(nth), nsims=8, maxsteps=2, theta_rtol=0.0, atol=1e-4, alpha=0.7)
    assert "MUSE_MAX_THETA" in str(e.value)
    with pytest.raises(M.MuseError) as e:
        prob.map_and_score_multi_async(5, 0, 4, np.zeros((2, nth)), atol=1e-4, z0_mode=0, result_area=1)
    assert "several maps" in str(e.value)
    res = M.muse(prob, np.zeros(nth), rng=3, nsims=24, maxsteps=6, get_covariance=True)
    ref = M.muse(OracleMuseProblem(x, model="funnel", ntheta=nth, prior=M.GaussianPrior(0.0, 3.0)), np.zeros(nth), rng=3, nsims=24,
                 maxsteps=6, get_covariance=True)
    np.testing.assert_allclose(res.theta, ref.theta, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(res.Sigma, ref.Sigma, rtol=1e-6, atol=1e-9)
    sig = np.sqrt(np.diag(res.Sigma))
    assert np.all(np.abs(res.theta - truth) < 5 * sig) and np.all(sig < 0.2)
    prob.close()
    with pytest.raises(M.MuseError) as e:
        M.HipMuseProblem(None, model="funnel", ntheta=65, N=1000)
    assert "MUSE_MAX_THETA_EXT" in str(e.value)
