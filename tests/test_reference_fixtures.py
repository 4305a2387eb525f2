"""The out-of-band pin of the oracle against the REAL MuseInference.jl (SURVEY.md §8 c: "parity unpinned").

tests/golden/reference_inputs.npz (committed; tests/golden/make_reference_inputs.py) holds the inputs;
julia/make_reference_fixtures.jl turns them into tests/golden/reference_outputs.npz with the reference package itself
(Optim LBFGS/HagerZhang through ẑ_at_θ, muse!/get_J!/get_H! with injected normals).  The build image has no julia, so
the outputs file is absent here and the comparison is reported as SKIPPED; with the file present it runs on CPU."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
INPUTS = os.path.join(HERE, "golden", "reference_inputs.npz")
OUTPUTS = os.path.join(HERE, "golden", "reference_outputs.npz")
MODELS = {0: "funnel", 1: "noise", 2: "smooth"}


def test_reference_inputs_are_current(O):
    """The committed inputs are what the generator writes today (same oracle streams)."""
    d = np.load(INPUTS)
    for c in range(int(d["ncases"])):
        model, N, th = MODELS[int(d[f"case{c}_model"])], int(d[f"case{c}_N"]), d[f"case{c}_theta"]
        x, z = O.sample_x_z(model, N, 7, c, th)
        assert np.array_equal(x, d[f"case{c}_x"])
    n1, n2 = O.normals(int(d["run_seed"]), 3, int(d["run_N"]))
    assert np.array_equal(n1, d["run_n1"][4]) and np.array_equal(n2, d["run_n2"][4])


@pytest.mark.skipif(not os.path.exists(OUTPUTS), reason="tests/golden/reference_outputs.npz absent: run "
                    "julia/make_reference_fixtures.jl where Julia and MuseInference.jl are installed")
def test_oracle_against_reference_outputs(O):
    """ẑ, Optim's iteration / evaluation counts and the muse! trajectory of the reference package against the oracle.
    The reference differentiates by ForwardDiff where the oracle uses closed forms: gradients agree to rounding, so
    counts are expected to be equal and ẑ to agree far inside the MAP tolerance; a differing count is reported with
    both paths' numbers (it can only come from a last-bit difference at a line-search decision)."""
    d, r = np.load(INPUTS), np.load(OUTPUTS)
    for c in range(int(d["ncases"])):
        model, th, atol = MODELS[int(d[f"case{c}_model"])], d[f"case{c}_theta"], float(d[f"case{c}_atol"])
        zo, io = O.zhat_at_theta(model, d[f"case{c}_x"], d[f"case{c}_z0"], th, atol)
        it, fc = int(r[f"case{c}_counts"][0]), int(r[f"case{c}_counts"][1])
        assert (io["iterations"], io["f_calls"]) == (it, fc), f"case {c}: oracle {io['iterations'], io['f_calls']} vs Optim {it, fc}"
        np.testing.assert_allclose(zo, r[f"case{c}_zhat"], rtol=0, atol=1e-8)
        np.testing.assert_allclose(-io["f_min"], -r[f"case{c}_fmin"][0], rtol=1e-10)
        np.testing.assert_allclose(O.grad_theta(model, d[f"case{c}_x"], zo, th), r[f"case{c}_score"], rtol=1e-8)
    # the muse! run: same algebra as tests/muse_reference.py on the oracle's map with the same normals
    import muse_reference as R
    from golden.make_golden import OracleMap, covariance, gaussian_prior  # noqa: F401  (generator helpers)
    pg, ph = gaussian_prior(float(d["run_prior_sigma"]))
    hist, theta, gs = R.muse_loop(OracleMap("funnel", d["run_x"], 1, int(d["run_seed"]), int(d["run_nsims"]), atol=float(d["run_atol"])),
                                  list(d["run_theta0"]), nsims=int(d["run_nsims"]), prior_grad_t=pg, prior_hess_t=ph,
                                  maxsteps=int(d["run_maxsteps"]), theta_rtol=float(d["run_theta_rtol"]), alpha=float(d["run_alpha"]))
    np.testing.assert_allclose(np.array([h["θ"] for h in hist]), r["run_thetas"], rtol=1e-7)
    np.testing.assert_allclose(np.array(theta), r["run_theta"], rtol=1e-6)
    np.testing.assert_allclose(np.array(gs), r["run_gs"], rtol=1e-7)
    J, Hs, H, step = covariance("funnel", d["run_x"], 1, int(d["run_seed"]), theta, gs, max(1, int(d["run_nsims"]) // 10),
                                atol=float(d["run_atol"]))
    np.testing.assert_allclose(np.array(J), r["run_J"], rtol=1e-6)
    np.testing.assert_allclose(np.array(H), r["run_H"], rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(OUTPUTS), reason="tests/golden/reference_outputs.npz absent: run "
                    "julia/make_reference_fixtures.jl where Julia and MuseInference.jl are installed")
def test_hip_against_reference_outputs(gpu, M):
    """The same comparison for the PRODUCT path: libmuse_hip's ẑ_at_θ (counts, ẑ, minimum, score) and a whole
    muse()/get_J!/get_H! run on HipMuseProblem against what the reference package produced from the same inputs (the
    normals injected into the reference are the engine's own Philox streams, bit for bit: test_sampler_bit_exact).  The
    day someone runs the Julia script, parity against the real package goes green (or red) in one step."""
    d, r = np.load(INPUTS), np.load(OUTPUTS)
    for c in range(int(d["ncases"])):
        model, th, atol = MODELS[int(d[f"case{c}_model"])], d[f"case{c}_theta"], float(d[f"case{c}_atol"])
        prob = M.HipMuseProblem(d[f"case{c}_x"], model=model, ntheta=th.size)
        zh, info = prob.zhat_at_theta(d[f"case{c}_x"], d[f"case{c}_z0"], th, atol)
        it, fc = int(r[f"case{c}_counts"][0]), int(r[f"case{c}_counts"][1])
        assert (info["iterations"], info["f_calls"]) == (it, fc), f"case {c}: HIP {info['iterations'], info['f_calls']} vs Optim {it, fc}"
        np.testing.assert_allclose(zh, r[f"case{c}_zhat"], rtol=0, atol=1e-8)
        np.testing.assert_allclose(-info["f_min"], -r[f"case{c}_fmin"][0], rtol=1e-10)
        np.testing.assert_allclose(prob.grad_theta_logLike(d[f"case{c}_x"], zh, th), r[f"case{c}_score"], rtol=1e-8)
        prob.close()
    prob = M.HipMuseProblem(d["run_x"], model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, float(d["run_prior_sigma"])))
    res = M.muse(prob, list(d["run_theta0"]), rng=int(d["run_seed"]), nsims=int(d["run_nsims"]), maxsteps=int(d["run_maxsteps"]),
                 theta_rtol=float(d["run_theta_rtol"]), grad_z_logLike_atol=float(d["run_atol"]), alpha=float(d["run_alpha"]),
                 get_covariance=True)
    np.testing.assert_allclose(np.array([h["θ"] for h in res.history]), r["run_thetas"], rtol=1e-7)
    np.testing.assert_allclose(res.theta, r["run_theta"], rtol=1e-6)
    np.testing.assert_allclose(np.array(res.gs), r["run_gs"], rtol=1e-7)
    np.testing.assert_allclose(res.J, np.atleast_2d(r["run_J"]), rtol=1e-6)
    np.testing.assert_allclose(res.H, np.atleast_2d(r["run_H"]), rtol=1e-4)
    prob.close()
