"""Model headers generated from the model's terms (museinference_jl_amd.symbolic; ElementwiseModel.from_expressions): where the
reference differentiates a SimpleMuseProblem's closures by AD (src/simple.jl:84-85), the derivatives a header has to state -- the
gradient, the score term, the second derivatives of the implicit-differentiation get_H! (src/muse.jl:335-405) -- are formed by
sympy from A(x, z), B(x, z) and the draw.

CPU: the generated text against the hand-written models/cubic.h through the CPU checker's build of both (draw, value, gradient,
score, MAP, implicit H), the contract refusals, per-element constants.  GPU: the generated model's engine library against the
checker's build of the same text, and get_H! by both branches.
"""
import os
import subprocess

import numpy as np
import pytest

from test_gpu_parity import assert_same_path_or_close

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CUBIC = os.path.join(ROOT, "museinference.jl_amd", "models", "cubic.h")

CUBIC_TERMS = dict(A="(x - (z + z**3/10))**2", B="z**2", z="sd*n1", x="z + z**3/10 + n2")
# a known spectrum P_i and a saturating (rational) response: x = z / sqrt(1 + z^2) + n2, z ~ N(0, e^theta P_i)
SAT_TERMS = dict(A="(x - z/sqrt(1 + z**2))**2", B="z**2/P", z="sd*sqrt(P)*n1", x="z/sqrt(1 + z**2) + n2")


def generated_cubic(M, directory=None):
    return M.ElementwiseModel.from_expressions("cubic_gen", directory=directory, **CUBIC_TERMS)


def spectrum(N):
    return 4.0 / (1.0 + np.arange(N) % 50) ** 1.2 + 0.1


# ------------------------------------------------------------------------------------------------ CPU
def test_generated_header_is_plain_c_and_states_what_it_came_from(M, tmp_path):
    m = generated_cubic(M, str(tmp_path))
    text = open(m.header).read()
    assert "GENERATED" in text and "#define MUSE_MODEL_SECOND 1" in text and '#define MUSE_MODEL_NAME "cubic_gen"' in text
    for fn in ("muse_model_sample", "muse_model_grad", "muse_model_score_term", "muse_model_second", "muse_model_dx_dsd"):
        assert text.count(f" {fn}(") == 1
    assert "pow(" not in text and "exp(" not in text
    subprocess.check_call(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Werror", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"),
                           "-include", "math.h", "-x", "c", m.header])
    assert generated_cubic(M, str(tmp_path)).header == m.header          # the same terms: the same text, the same library
    assert not m.pair and m.runtime_constants == []


def test_generated_cubic_equals_the_hand_written_header_on_the_oracle(M, O, tmp_path):
    """The same model twice -- terms differentiated by sympy, and models/cubic.h written by hand -- in the CPU checker: equal
    draws, values, gradients, scores, MAPs and implicit-differentiation H to rounding (the two evaluate different but
    equivalent IEEE sequences)."""
    m = generated_cubic(M, str(tmp_path))
    N, theta = 1201, [0.4, -0.3, 0.9]
    got = {}
    for tag, (hdr, name) in {"gen": (m.header, m.library_name), "hand": (CUBIC, "cubic")}.items():
        with O.user_model(hdr, name):
            x, z = O.sample_x_z("user", N, 17, 3, theta)
            zz = 0.6 * z + 0.05
            f, g = O.logLike_and_grad_z("user", x, zz, theta)
            s = O.grad_theta("user", x, zz, theta)
            zh, info = O.zhat_at_theta("user", x, np.zeros(N), theta, 1e-8)
            H, _ = O.implicit_H("user", N, 5, 0, theta, atol=1e-12)
            got[tag] = dict(x=x, z=z, f=f, g=g, s=s, zh=zh, H=H, status=info["status"])
    a, b = got["gen"], got["hand"]
    # (2 = the value stopped changing, Optim's f_converged with f_tol = 0: at |g| ~ 3e-7 a step lowers f ~ 880 by less than an
    #  ulp; which of the two stops a header meets first depends on its rounding)
    assert a["status"] in (0, 2) and b["status"] in (0, 2)
    assert np.array_equal(a["z"], b["z"])
    np.testing.assert_allclose(a["x"], b["x"], rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(a["f"], b["f"], rtol=1e-13)
    np.testing.assert_allclose(a["g"], b["g"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(a["s"], b["s"], rtol=1e-12)
    np.testing.assert_allclose(a["zh"], b["zh"], atol=2e-6)
    np.testing.assert_allclose(a["H"], b["H"], rtol=1e-8, atol=1e-8 * np.abs(b["H"]).max())


def test_generated_model_passes_the_consistency_checks(M, O):
    """check_model_consistency (first derivatives against differences of the value) through an oracle-backed problem, and the
    generated second derivatives against differences of the generated first ones on the host (muse_model_eval of the model's own
    engine library -- the one __graft_entry__.build() pre-builds -- needs no GPU for a model without run-time constants)."""
    from oracle_problem import OracleMuseProblem
    from museinference_jl_amd import models as MM
    m = generated_cubic(M)
    with O.user_model(m.header, m.library_name):
        res = M.check_model_consistency(OracleMuseProblem(None, model="user", ntheta=3, N=2001), [0.4, -0.3, 0.9], rng=5)
    assert res["grad_z"] <= 2e-5 + res["noise_floor"] and res["grad_theta"] <= 2e-5 + res["noise_floor"]
    lib = M._capi.load_library(m.library())
    assert lib.muse_model_has_second() == 1

    def ev(iv, sd, x, z, n1, n2, i=0):
        out = np.empty(12)
        assert lib.muse_model_eval(None, iv, sd, x, z, n1, n2, int(i), M._capi.ptr(out)) == 0
        return dict(zip(("grad", "term", "B", "ozz", "ozx", "bz", "bx", "z", "x", "dx_dsd"), out.tolist()))
    rs = np.random.RandomState(2)
    assert MM._check_second(ev, np.array([0.4, -0.3]), rs.randn(600), 0.7 * rs.randn(600), 8, 2e-5) <= 2e-5
    e = ev(0.7, 1.2, 0.9, 0.4, -0.3, 0.8)                      # ... and against models/cubic.h's closed forms at one point
    hp, r = 1 + 0.3 * 0.16, 0.9 - (0.4 + 0.1 * 0.4 ** 3)
    np.testing.assert_allclose([e["grad"], e["B"], e["ozz"], e["ozx"], e["bz"], e["bx"]],
                               [0.7 * 0.4 - r * hp, 0.16, 0.7 + hp * hp - r * 0.24, -hp, 0.8, 0.0], rtol=1e-13, atol=1e-300)


def test_terms_with_per_element_constants_and_a_square_root(M, O, tmp_path):
    N, theta = 801, [0.3, -0.5]
    P = spectrum(N)
    m = M.ElementwiseModel.from_expressions("saturating_gen", directory=str(tmp_path), constants={"P": P}, **SAT_TERMS)
    text = open(m.header).read()
    assert "P(i)" in text and "sqrt(" in text and f"#define MUSE_MODEL_N {N}" in text
    from oracle_problem import OracleMuseProblem
    with O.user_model(m.header, m.library_name):
        x, z = O.sample_x_z("user", N, 4, 1, theta)
        n1, n2 = O.normals(4, 1, N)
        k = (np.arange(N) * 2) // N
        np.testing.assert_allclose(z, np.exp(0.5 * np.asarray(theta))[k] * np.sqrt(P) * n1, rtol=1e-14)
        np.testing.assert_allclose(x, z / np.sqrt(1 + z * z) + n2, rtol=1e-14, atol=1e-15)
        zz = 0.7 * z - 0.02
        f, g = O.logLike_and_grad_z("user", x, zz, theta)
        hz = zz / np.sqrt(1 + zz * zz)
        want = -0.5 * np.sum((x - hz) ** 2 + np.exp(-np.asarray(theta))[k] * zz * zz / P) - 0.5 * np.sum(np.asarray(theta)[k])
        np.testing.assert_allclose(f, want, rtol=1e-13)
        np.testing.assert_allclose(g, (x - hz) * (1 + zz * zz) ** -1.5 - np.exp(-np.asarray(theta))[k] * zz / P, rtol=1e-11, atol=1e-13)
        res = M.check_model_consistency(OracleMuseProblem(None, model="user", ntheta=2, N=N), theta, rng=3)
        assert max(res["grad_z"], res["grad_theta"]) <= 2e-5 + res["noise_floor"]
        H, _ = O.implicit_H("user", N, 5, 0, theta, atol=1e-12)
        _, zfid, _ = O.map_and_score_batch("user", N, 5, 0, 1, theta, atol=1e-12, z0_mode=0)
        Hfd = O.fd_jacobian("user", N, 5, 0, theta, [1e-3] * 2, zfid[0], atol=1e-12)
        np.testing.assert_allclose(H, Hfd, rtol=2e-5, atol=2e-5 * np.abs(Hfd).max())


def test_terms_outside_the_family_are_refused(M, tmp_path):
    """What the header contract forbids is refused when the text is generated, with the reason: a contribution of the pad element
    (x = z = 0), a transcendental function (host, device and checker must evaluate one IEEE sequence), symbols A and B may not
    depend on, a name that is not an identifier."""
    f = lambda **kw: M.ElementwiseModel.from_expressions(kw.pop("name", "bad"), directory=str(tmp_path), **{**CUBIC_TERMS, **kw})
    with pytest.raises(ValueError, match="vanish at x = z = 0"):
        f(A="(x - z - 1)**2")
    with pytest.raises(ValueError, match="vanish at x = z = 0"):
        f(B="z**2 + 1")
    with pytest.raises(ValueError, match="not allowed in a model header"):
        f(A="(x - sin(z))**2", x="sin(z) + n2")
    with pytest.raises(ValueError, match="half-integer exponents"):
        f(B="z**2 * (1 + z**2)**(1/3)")
    with pytest.raises(ValueError, match="functions of x and z"):
        f(A="(x - z*sd)**2")
    with pytest.raises(ValueError, match="function of sd, n1, n2"):
        f(z="sd*n1 + x")
    with pytest.raises(ValueError, match="model name"):
        f(name="../evil")
    assert os.listdir(str(tmp_path)) == []           # nothing was written for a refused model


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("N,nth,theta", [(3000, 2, [0.4, -0.2]), (10000, 1, [0.3]), (9999, 3, [0.5, 0.0, -0.4]), (70001, 2, [0.3, 0.1])])
def test_generated_model_hip_against_oracle(gpu, M, O, N, nth, theta):
    """The generated header's engine library against the CPU checker's build of the same text: draws bit-equal, the batched map
    on the same solver path (or within the documented tolerance), and the implicit-differentiation H from the generated second
    derivatives."""
    m = generated_cubic(M)
    prob = M.HipMuseProblem(None, model=m, ntheta=nth, N=N)
    assert prob.has_second_derivatives
    with O.user_model(m.header, m.library_name):
        x, z = prob.sample_x_z(M.SimRng(11, 2), theta)
        xo, zo = O.sample_x_z("user", N, 11, 2, theta)
        assert np.array_equal(x, xo) and np.array_equal(z, zo)
        g, info = prob.map_and_score_batch(11, 0, 4, theta, atol=1e-4, z0_mode=0)
        zh = prob.get_zhat(0, 4)
        go, zho, io = O.map_and_score_batch("user", N, 11, 0, 4, theta, atol=1e-4, z0_mode=0)
        long = int(io["iterations"].max()) > 20
        same = assert_same_path_or_close(info, io, zh, zho, g, go, 1e-4, theta, "funnel", z_atol=1e-7 if long else 1e-9, g_rtol=1e-6 if long else 1e-10)
        assert same.all() if not long else same.mean() >= 0.5
        Hs, its = prob.implicit_H_batch(5, 0, 2, theta, atol=1e-1, cg_maxiter=100)
        for s in range(2):
            Ho, ito = O.implicit_H("user", N, 5, s, theta, atol=1e-1, cg_maxiter=100)
            assert np.all(np.abs(its[s] - ito) <= 1)
            np.testing.assert_allclose(Hs[s], Ho, rtol=1e-7, atol=1e-7 * np.abs(Ho).max())
    res = M.check_model_consistency(prob, theta, rng=4)          # first AND second derivatives (the header has MUSE_MODEL_SECOND)
    assert max(res["grad_z"], res["grad_theta"]) <= 2e-5 + res["noise_floor"] and res["second"] <= 2e-5
    prob.close()


@pytest.mark.gpu
def test_generated_model_whole_run_equals_the_hand_written_model(gpu, M, O):
    """muse() with covariance on the generated model and on models/cubic.h, same data and streams: the same estimate to the
    solver's tolerance (equivalent arithmetic, different rounding), J and H by both get_H! branches included."""
    with O.user_model(CUBIC, "cubic"):
        x, _ = O.sample_x_z("user", 4000, 9, M.DATA_SIM, [0.2, -0.3])
    out = []
    for model in (generated_cubic(M), M.ElementwiseModel.packaged("cubic")):
        prob = M.HipMuseProblem(x, model=model, ntheta=2, prior=M.GaussianPrior(0.0, 3.0))
        res = M.muse(prob, [0.0, 0.0], rng=2, nsims=40, maxsteps=8, get_covariance=True)
        Hfd = res.H.copy()
        res.Hs, res.H = [], None
        M.get_H_(res, prob, nsims=8, implicit_diff=True)
        out.append((res.theta.copy(), res.J.copy(), Hfd, res.H.copy()))
        prob.close()
    (ta, Ja, Ha, Ia), (tb, Jb, Hb, Ib) = out
    np.testing.assert_allclose(ta, tb, atol=1e-5)
    np.testing.assert_allclose(Ja, Jb, rtol=1e-4, atol=1e-4 * np.abs(Jb).max())
    np.testing.assert_allclose(Ha, Hb, rtol=1e-3, atol=1e-3 * np.abs(Hb).max())
    np.testing.assert_allclose(Ia, Ib, rtol=1e-4, atol=1e-4 * np.abs(Ib).max())


# ================================================================================================ the two-parameter family
NMV_TERMS = dict(coefs=["a", "exp(b/2)", "exp(-b)"], C="b", o="(x - z)**2 + c2*(z - c0)**2", z="c0 + c1*n1", x="z + n2")


def generated_nmv(M, directory=None):
    return M.ElementwiseModel.from_pair_expressions("nmv_gen", directory=directory, **NMV_TERMS)


def test_generated_pair_header_on_the_checker_against_closed_forms(M, O, tmp_path):
    """z_i ~ N(mu_k, e^tau_k), x_i ~ N(z_i, 1) from its terms (models/normal_mean_var.h is the hand-written one): the checker's build
    of the generated text against the model's closed forms -- draw, value, gradient, BOTH kinds of score component, the MAP --, the
    hand-written header's draw bit for bit, and the consistency check AD would have made unnecessary."""
    from oracle_problem import OracleMuseProblem
    from test_pair_model import HEADER, NAME, blocks, closed_forms
    m = generated_nmv(M, str(tmp_path))
    text = open(m.header).read()
    assert m.pair and "#define MUSE_MODEL_PAIR 1" in text and text.count("muse_model_exp(") == 2 and " exp(" not in text.split("*/")[1]
    subprocess.check_call(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Werror", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"),
                           "-include", "math.h", "-x", "c", m.header])
    N, theta = 1001, np.array([0.7, -0.3, 0.4, 1.1])
    K = 2
    k = blocks(N, K)
    with O.user_model(HEADER, NAME):
        xh, zh_ = O.sample_x_z("user", N, 5, 3, theta)
    with O.user_model(m.header, m.library_name):
        x, z = O.sample_x_z("user", N, 5, 3, theta)
        np.testing.assert_allclose(z, zh_, rtol=1e-14, atol=1e-15)        # (c0 + c1 n1 against fma(c1, n1, c0))
        np.testing.assert_allclose(x, xh, rtol=1e-14, atol=1e-15)
        zz = 0.7 * z + 0.1
        f, g = O.logLike_and_grad_z("user", x, zz, theta)
        fo, go, so = closed_forms(x, zz, theta)
        np.testing.assert_allclose(f, fo, rtol=1e-13)
        np.testing.assert_allclose(g, go, rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(O.grad_theta("user", x, zz, theta), so, rtol=1e-12, atol=1e-12)
        zhat, info = O.zhat_at_theta("user", x, np.zeros(N), theta, atol=1e-9)
        iv = np.exp(-theta[K + k])
        np.testing.assert_allclose(zhat, (x + iv * theta[k]) / (1 + iv), rtol=0, atol=1e-9)
        res = M.check_model_consistency(OracleMuseProblem(None, model="user", ntheta=4, N=2001), [0.4, -0.3, 0.9, 0.2], rng=5)
    assert res["grad_z"] <= 2e-5 + res["noise_floor"] and res["grad_theta"] <= 2e-5 + res["noise_floor"]


def test_pair_terms_outside_the_family_are_refused(M, tmp_path):
    f = lambda **kw: M.ElementwiseModel.from_pair_expressions("bad", directory=str(tmp_path), **{**NMV_TERMS, **kw})
    with pytest.raises(ValueError, match="vanish at x = z = 0 with zero coefficients"):
        f(o="(x - z)**2 + (z - c0)**2/c1**2")                     # a coefficient in a denominator: the pad element gives 0/0
    with pytest.raises(ValueError, match="vanish at x = z = 0 with zero coefficients|not a function of the coefficients alone"):
        f(coefs=["a", "exp(b/2)", "a*exp(-b)"], o="(x - z)**2 + c2*z")   # d c2 / d a = e^-b, which no coefficient holds
    with pytest.raises(ValueError, match="may read c0 and c1 only"):
        f(z="c0 + n1/sqrt(c2)")
    with pytest.raises(ValueError, match="between one and four"):
        f(coefs=["a", "b", "a*b", "a + b", "a - b"])
    with pytest.raises(ValueError, match="not allowed in a model header"):
        f(o="(x - z)**2 + c2*(z - c0)**2 + z*sin(z)")
    assert os.listdir(str(tmp_path)) == []


@pytest.mark.gpu
@pytest.mark.parametrize("N,nth,theta,split", [(300, 4, [0.4, -0.3, 0.9, 0.1], 0), (10000, 2, [1.0, 1.0], 0), (9001, 6, [0.0, 0.7, -0.4, 0.3, 0.0, -0.3], 8),
                                               (70001, 4, [0.3, -0.2, 0.1, 0.9], 0)])
def test_generated_pair_model_hip_against_the_checker(gpu, M, O, N, nth, theta, split):
    """The generated two-parameter header's engine library against the checker's build of the same text (resident, LDS-resident,
    register clusters, streaming clusters): the draw bit for bit, maps on the same solver path with scores rtol 1e-10, the MAP in
    closed form, the finite-difference get_H!."""
    from test_pair_model import blocks
    theta = np.asarray(theta)
    m = generated_nmv(M)
    K = nth // 2
    k = blocks(N, K)
    with O.user_model(m.header, m.library_name):
        xdata, _ = O.sample_x_z("user", N, 77, M.DATA_SIM, np.concatenate([np.full(K, 0.3), np.zeros(K)]))
        prob = M.HipMuseProblem(xdata, model=m, ntheta=nth)
        if split:
            prob.set_element_split(split)
        x, z = prob.sample_x_z(M.SimRng(1234, 2**40 + 7), theta)
        xo, zo = O.sample_x_z("user", N, 1234, 2**40 + 7, theta)
        assert np.array_equal(z, zo) and np.array_equal(x, xo)
        nsims = 5 if N > 20000 else 13
        g, info = prob.map_and_score_batch(42, 3, 3 + nsims, theta, include_data=True, atol=1e-6, z0_mode=0)
        go, zo, io = O.map_and_score_batch("user", N, 42, 3, 3 + nsims, theta, atol=1e-6, x_data=xdata, z0_mode=0)
        zh = prob.get_zhat(0, nsims + 1)
        same = assert_same_path_or_close(info, io, zh, zo, g, go, 1e-6, theta, "funnel", g_rtol=1e-10)
        assert same.all() and np.all(info["status"] == 0)
        iv = np.exp(-theta[K + k])
        np.testing.assert_allclose(zh[0], (xdata + iv * theta[k]) / (1 + iv), rtol=0, atol=2e-6)
        step = np.full(nth, 0.05)
        Hs, _ = prob.fd_jacobian_batch(11, 2, 4, theta, step, atol=1e-6)
        _, zfid, _ = O.map_and_score_batch("user", N, 11, M.MASTER_SIM, M.MASTER_SIM + 1, theta, atol=1e-6, z0_mode=0)
        for s in range(2):
            Ho = O.fd_jacobian("user", N, 11, 2 + s, theta, step, zfid[0], atol=1e-6)
            np.testing.assert_allclose(Hs[s], Ho, rtol=1e-7, atol=1e-7 * np.abs(Ho).max())
    res = M.check_model_consistency(prob, theta, rng=4)
    assert max(res["grad_z"], res["grad_theta"]) <= 2e-5 + res["noise_floor"]
    prob.close()


@pytest.mark.gpu
def test_muse_on_the_generated_pair_model_against_the_exact_posterior(gpu, M):
    """muse() with covariance on the generated model, device-resident loop and host loop the same bits, against the exact marginal
    posterior of x_i ~ N(mu_k, 1 + e^tau_k) (tests/test_pair_model.py's acceptance test for the hand-written header)."""
    from test_pair_model import PRIOR_SIGMA, exact_posterior
    m = generated_nmv(M)
    N, K, truth, nsims = 10000, 2, [0.8, -0.6, 0.5, 1.0], 256
    tmp = M.HipMuseProblem(None, model=m, ntheta=2 * K, N=N)
    x, _ = tmp.sample_x_z(M.SimRng(99, M.DATA_SIM), truth)
    tmp.close()
    mode, sigma = exact_posterior(x, K)
    prob = M.HipMuseProblem(x, model=m, ntheta=2 * K, prior=M.GaussianPrior(0.0, PRIOR_SIGMA))
    kw = dict(nsims=40, maxsteps=5, theta_rtol=0.0, atol=1e-6, alpha=0.7)
    a, b = prob.run_muse(3, [0.0] * 4, device_loop=True, **kw), prob.run_muse(3, [0.0] * 4, device_loop=False, **kw)
    assert a[0] == b[0] == 5 and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
    res = M.muse(prob, [0.0] * (2 * K), rng=20240, nsims=nsims, maxsteps=60, theta_rtol=1e-5, grad_z_logLike_atol=1e-6, alpha=1.0, get_covariance=True)
    dev = np.abs(np.asarray(res.theta) - mode) / (sigma / np.sqrt(nsims))
    assert np.all(dev < 4.0), (res.theta, mode, dev)
    got = np.sqrt(np.diag(np.atleast_2d(res.Sigma)))
    assert np.all(np.abs(got / sigma - 1.0) < 5.0 * 0.5 * np.sqrt(2.0 / (nsims - 1)) + 0.03), (got, sigma)
    prob.close()


HEAVY = os.path.join(HERE, "models", "pair_heavy_score.h")


@pytest.mark.gpu
def test_a_loop_kernel_beyond_the_scratch_bound_is_not_launched(gpu, M, capfd):
    """A correct header can cost a loop kernel most of its registers (tests/models/pair_heavy_score.h: 420 bytes of scratch per lane
    for three or four blocks in the LDS-resident placement).  Beyond the product's bound (256 bytes) muse_run_device runs the host
    loop -- the same bits, solver records included -- while the same header's kernels within the bound still run as one launch."""
    model = M.ElementwiseModel("pair_heavy_score", HEAVY)
    for N, nth, th0, device in ((10000, 8, [0.2, -0.1, 0.3, 0.0, 0.5, -0.5, 0.0, 1.0], False), (10000, 2, [0.1, 0.4], True),
                                (3000, 8, [0.0] * 8, True)):
        x = np.sin(0.3 * np.arange(N)) + 0.4 + 0.8 * np.cos(1.7 * np.arange(N))
        prob = M.HipMuseProblem(x, model=model, ntheta=nth, prior=M.GaussianPrior(0.0, 3.0))
        kw = dict(nsims=40, maxsteps=4, theta_rtol=0.0, atol=1e-6, alpha=0.7)
        b = prob.run_muse(3, th0, device_loop=False, **kw)
        capfd.readouterr()
        prob.debug_flags(M.HipMuseProblem.DEBUG_RUN_TIMING)
        a = prob.run_muse(3, th0, device_loop=True, **kw)
        prob.debug_flags(0)
        assert ("[muse_run_device] launch call" in capfd.readouterr().err) == device, (N, nth)
        assert a[0] == b[0] == 4 and np.array_equal(a[1], b[1]) and np.array_equal(a[2][:, :-1], b[2][:, :-1]) and np.array_equal(a[3], b[3])
        assert a[4].tobytes() == b[4].tobytes(), (N, nth)          # iterations, evaluation counts, history words, status, f, |g|
        prob.close()
    # the generated header of the same model keeps the factor on the sums (symbolic.py, split): its loop kernels are within the bound
    gen = generated_nmv(M)
    x = np.sin(0.3 * np.arange(10000)) + 0.4
    prob = M.HipMuseProblem(x, model=gen, ntheta=8, prior=M.GaussianPrior(0.0, 3.0))
    kw = dict(nsims=40, maxsteps=4, theta_rtol=0.0, atol=1e-6, alpha=0.7)
    b = prob.run_muse(3, [0.1] * 8, device_loop=False, **kw)
    capfd.readouterr()
    prob.debug_flags(M.HipMuseProblem.DEBUG_RUN_TIMING)
    a = prob.run_muse(3, [0.1] * 8, device_loop=True, **kw)
    prob.debug_flags(0)
    assert "[muse_run_device] launch call" in capfd.readouterr().err
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3]) and a[4].tobytes() == b[4].tobytes()
    prob.close()
