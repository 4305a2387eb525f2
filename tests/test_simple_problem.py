"""The general front-end: a problem given as closures and differentiated by AD (the reference's SimpleMuseProblem, src/simple.jl:4-12,
79-95) -- simple.TorchMuseProblem -- and the interface's default ẑ_at_θ behind it (src/interface.jl:140-166; optim.py).

CPU (torch on the host): the L-BFGS/HagerZhang of optim.py against the CPU checker's restatement of the same published algorithms on
the funnel and cubic objectives -- the same iteration and evaluation counts --; the funnel written as closures through muse() against
the exact marginal posterior; a model NO header holds (a dense mixing matrix: every latent variable in every observation) against its
closed-form posterior; transforms.  GPU: the same problems with the tensors on the device."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


def funnel_closures(N, nth):
    k = (torch.arange(N) * nth) // N

    def sample_x_z(gen, theta):
        kk = k.to(theta.device)
        z = torch.exp(theta[kk] / 2) * torch.randn(N, generator=gen, device=theta.device, dtype=theta.dtype)
        return z + torch.randn(N, generator=gen, device=theta.device, dtype=theta.dtype), z

    def logLike(x, z, theta):
        kk = k.to(z.device)
        return -0.5 * (torch.sum((x - z) ** 2) + torch.sum(torch.exp(-theta[kk]) * z ** 2) + torch.sum(theta[kk]))
    return sample_x_z, logLike


def gaussian_prior(sigma):
    return lambda theta: -0.5 * torch.sum(theta ** 2) / sigma ** 2


# ------------------------------------------------------------------------------------------------ the solver
@pytest.mark.parametrize("model,N,theta,atol", [("funnel", 700, [0.3, -0.8], 1e-2), ("funnel", 1500, [1.0], 1e-8), ("funnel", 64, [-2.0, 0.5, 1.5], 1e-6)])
def test_lbfgs_hagerzhang_takes_the_checkers_path_on_the_funnel(M, O, model, N, theta, atol):
    """optim.lbfgs on the funnel's objective (closed-form gradient, torch float64 on the host) against the checker's zhat_at_theta:
    equal iteration and evaluation counts and status, the MAP to rounding (the two sum in different orders)."""
    from museinference_jl_amd import optim
    nth = len(theta)
    k = torch.as_tensor((np.arange(N) * nth) // N)
    iv = torch.as_tensor(np.exp(-np.asarray(theta)))[k]
    for sim in range(4):
        x, z = O.sample_x_z(model, N, 7, sim, theta)
        xt = torch.as_tensor(x)

        def fg(zz):
            r = xt - zz
            return 0.5 * (torch.sum(r * r) + torch.sum(iv * zz * zz) + float(np.sum(np.asarray(theta)[k.numpy()]))), -(r - iv * zz)
        for z0 in (np.zeros(N), z):
            zh, info = optim.lbfgs(fg, torch.as_tensor(z0), atol)
            zo, io = O.zhat_at_theta(model, x, z0, theta, atol)
            assert (info["iterations"], info["f_calls"], info["status"]) == (io["iterations"], io["f_calls"], io["status"]), (sim, info, io)
            np.testing.assert_allclose(zh.numpy(), zo, rtol=0, atol=1e-9)
            np.testing.assert_allclose(info["f_min"], io["f_min"], rtol=1e-12)


def test_lbfgs_hagerzhang_on_a_non_quadratic_objective(M, O):
    """The cubic user model's objective (tens of iterations with real line searches and a wrapping history): the checker's build of
    models/cubic.h against optim.lbfgs on the same function written in torch -- both converge to the same MAP; where the two take the
    same path (every committed case) the counts are equal."""
    import os
    from museinference_jl_amd import optim
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "museinference.jl_amd", "models", "cubic.h")
    N, theta = 400, [0.4, -0.3]
    k = torch.as_tensor((np.arange(N) * 2) // N)
    iv = torch.as_tensor(np.exp(-np.asarray(theta)))[k]
    h = lambda v: v + 0.1 * v ** 3
    with O.user_model(hdr, "cubic"):
        for sim in range(3):
            x, _ = O.sample_x_z("user", N, 17, sim, theta)
            xt = torch.as_tensor(x)

            def fg(zz):
                r = xt - h(zz)
                return 0.5 * (torch.sum(r * r) + torch.sum(iv * zz * zz) + float(np.sum(np.asarray(theta)[k.numpy()]))), iv * zz - r * (1 + 0.3 * zz * zz)
            zh, info = optim.lbfgs(fg, torch.zeros(N, dtype=torch.float64), 1e-6)
            zo, io = O.zhat_at_theta("user", x, np.zeros(N), theta, 1e-6)
            assert info["status"] == io["status"] == 0 and info["iterations"] > 5
            np.testing.assert_allclose(zh.numpy(), zo, atol=5e-6)
            assert abs(info["iterations"] - io["iterations"]) <= 2 and abs(info["f_calls"] - io["f_calls"]) <= 6, (info, io)


def test_solver_edge_cases():
    from museinference_jl_amd import optim
    quad = lambda z: (0.5 * torch.sum(z * z), z.clone())
    z, info = optim.lbfgs(quad, torch.zeros(5, dtype=torch.float64), 1e-8)          # converged at the start
    assert info["iterations"] == 0 and info["f_calls"] == 1 and info["status"] == optim.STATUS_G_CONVERGED
    z, info = optim.lbfgs(quad, torch.full((5,), 3.0, dtype=torch.float64), 1e-12)
    assert info["status"] == optim.STATUS_G_CONVERGED and info["iterations"] == 1 and float(z.abs().max()) == 0.0
    nan = lambda z: (torch.tensor(float("nan")), z.clone())
    z, info = optim.lbfgs(nan, torch.ones(3, dtype=torch.float64), 1e-8)
    assert info["status"] == optim.STATUS_NONFINITE and info["iterations"] == 0
    rosen = lambda z: ((1 - z[0]) ** 2 + 100 * (z[1] - z[0] ** 2) ** 2,
                       torch.stack([-2 * (1 - z[0]) - 400 * z[0] * (z[1] - z[0] ** 2), 200 * (z[1] - z[0] ** 2)]))
    # Optim.jl's documented run of `optimize(f, g!, [0.0, 0.0], LBFGS())` on this function prints Iterations: 24, f(x) calls: 67
    # (tests/test_oracle.py holds the checker to the same two counters; quoted from memory, see there)
    z, info = optim.lbfgs(rosen, torch.tensor([0.0, 0.0], dtype=torch.float64), 1e-8)
    assert (info["iterations"], info["f_calls"], info["status"]) == (24, 67, optim.STATUS_G_CONVERGED)
    z, info = optim.lbfgs(rosen, torch.tensor([-1.2, 1.0], dtype=torch.float64), 1e-10)
    assert info["status"] == optim.STATUS_G_CONVERGED and 10 < info["iterations"] < 60
    np.testing.assert_allclose(z.numpy(), [1.0, 1.0], atol=1e-8)


# ------------------------------------------------------------------------------------------------ the problem
def run_funnel(M, device, N=2000, nsims=60):
    from test_exact_marginal import exact_scale_family
    sample_x_z, logLike = funnel_closures(N, 1)
    sim = M.TorchMuseProblem(None, sample_x_z, logLike, device=device)
    x, _ = sim.sample_x_z(M.SimRng(5, M.DATA_SIM), [0.7])
    prob = M.TorchMuseProblem(x, sample_x_z, logLike, logPrior=gaussian_prior(3.0), device=device)
    # the operators: autograd against the closed forms
    xs, zs = prob.sample_x_z(M.SimRng(5, 3), [0.7])
    xs2, _ = prob.sample_x_z(M.SimRng(5, 3), [0.7])
    assert torch.equal(xs, xs2) and xs.device.type == torch.device(device).type
    f, g = prob.logLike_and_grad_z_logLike(xs, 0.5 * zs, [0.7])
    np.testing.assert_allclose(g.cpu().numpy(), ((xs - 0.5 * zs) - np.exp(-0.7) * 0.5 * zs).cpu().numpy(), rtol=1e-12, atol=1e-13)
    s = prob.grad_theta_logLike(xs, zs, [0.7])
    np.testing.assert_allclose(s, [0.5 * (np.exp(-0.7) * float(torch.sum(zs ** 2)) - N)], rtol=1e-12)
    zh, info = prob.zhat_at_theta(xs, torch.zeros_like(xs), [0.7], 1e-8)
    np.testing.assert_allclose(zh.cpu().numpy(), (xs / (1 + np.exp(-0.7))).cpu().numpy(), atol=1e-8)
    assert int(info["status"]) == 0 and int(info["iterations"]) == 1 and int(info["f_calls"]) == 3       # the isotropic problem's path
    np.testing.assert_allclose(prob.grad_logPrior_theta([0.6]), [-0.6 / 9.0], rtol=1e-12)
    np.testing.assert_allclose(prob.hess_logPrior_theta([0.6]), [[-1.0 / 9.0]], rtol=1e-12)
    res = M.muse(prob, [0.0], rng=11, nsims=nsims, maxsteps=30, theta_rtol=1e-4, grad_z_logLike_atol=1e-6, alpha=1.0, get_covariance=True)
    mode, sigma = exact_scale_family(x.cpu().numpy(), 1)
    assert np.all(np.abs(res.theta - mode) / (sigma / np.sqrt(nsims)) < 4.0), (res.theta, mode, sigma)
    assert np.all(np.abs(np.sqrt(np.diag(np.atleast_2d(res.Sigma))) / sigma - 1.0) < 0.6)
    return res


def test_funnel_as_closures_against_the_exact_posterior(M):
    run_funnel(M, "cpu")
    # ... and in single precision (the reference's problems may be Float32, src/turing.jl:188; the HIP engine is fp64 only)
    from test_exact_marginal import exact_scale_family
    sample_x_z, logLike = funnel_closures(2000, 1)
    x, _ = M.TorchMuseProblem(None, sample_x_z, logLike, dtype=torch.float32).sample_x_z(M.SimRng(5, M.DATA_SIM), [0.7])
    prob = M.TorchMuseProblem(x, sample_x_z, logLike, logPrior=gaussian_prior(3.0), dtype=torch.float32)
    res = M.muse(prob, [0.0], rng=11, nsims=40, maxsteps=20, theta_rtol=1e-3, grad_z_logLike_atol=1e-3, get_covariance=True)
    mode, sigma = exact_scale_family(x.double().numpy(), 1)
    assert x.dtype == torch.float32 and abs(res.theta[0] - mode[0]) < 4.0 * sigma[0] / np.sqrt(40) + 0.01


def mixing_problem(M, device, n=24, m=40, seed=2):
    """z ~ N(0, e^theta I_n), x ~ N(A z, I_m) with a dense m x n matrix A: every latent variable in every observation -- a logLike no
    elementwise header holds.  Marginally x ~ N(0, I + e^theta A A'): the exact posterior of theta is one-dimensional quadrature."""
    rs = np.random.RandomState(seed)
    A = torch.as_tensor(rs.randn(m, n) / np.sqrt(n), device=device)

    def sample_x_z(gen, theta):
        z = torch.exp(theta[0] / 2) * torch.randn(n, generator=gen, device=theta.device, dtype=theta.dtype)
        return A.to(theta.device) @ z + torch.randn(m, generator=gen, device=theta.device, dtype=theta.dtype), z

    def logLike(x, z, theta):
        r = x - A.to(z.device) @ z
        return -0.5 * (torch.sum(r * r) + torch.exp(-theta[0]) * torch.sum(z * z) + n * theta[0])
    return A, sample_x_z, logLike


def run_mixing(M, device, nrep=12):
    nsims = 50
    A, sample_x_z, logLike = mixing_problem(M, device)
    An = A.cpu().numpy()
    lam = np.linalg.eigvalsh(An @ An.T)
    sim = M.TorchMuseProblem(None, sample_x_z, logLike, device=device)
    # several independent data sets at once (block structure over the replicas would be a header's job; here: one problem per replica)
    devs = []
    for rep in range(nrep):
        x, _ = sim.sample_x_z(M.SimRng(100 + rep, M.DATA_SIM), [0.4])
        prob = M.TorchMuseProblem(x, sample_x_z, logLike, logPrior=gaussian_prior(3.0), device=device)
        res = M.muse(prob, [0.0], rng=rep, nsims=nsims, maxsteps=30, theta_rtol=1e-3, grad_z_logLike_atol=1e-6, alpha=1.0, get_covariance=True)
        U = np.linalg.eigh(An @ An.T)[1]
        y2 = (U.T @ x.cpu().numpy()) ** 2
        ts = np.linspace(-6, 6, 4001)
        lp = np.array([-0.5 * np.sum(np.log1p(np.exp(t) * lam) + y2 / (1 + np.exp(t) * lam)) - 0.5 * t * t / 9.0 for t in ts])
        w = np.exp(lp - lp.max())
        w /= w.sum()
        mean = np.sum(w * ts)
        sd = np.sqrt(np.sum(w * (ts - mean) ** 2))
        devs.append((float(res.theta[0]) - ts[np.argmax(lp)]) / sd)
        assert 0.3 < float(np.sqrt(res.Sigma[0, 0])) / sd < 3.0
    # MUSE is not exact for this non-isotropic model, but it is asymptotically unbiased: over the replicas its estimate sits within the
    # posterior's width of the exact mode, without a systematic offset
    devs = np.array(devs)
    assert np.all(np.abs(devs) < 2.5) and abs(devs.mean()) < (1.0 if nrep >= 10 else 1.5), devs


def test_a_model_no_header_holds_against_its_exact_posterior(M):
    run_mixing(M, "cpu")


def test_transformed_theta_through_the_closure_problem(M):
    """A problem in a transformed space (theta' = log sigma^2 handled by a subclass's transform pair): check_self_consistency and the
    chain rule of grad_theta_logLike in the transformed space."""
    N = 300
    k = 1

    def sample_x_z(gen, theta):       # theta = sigma^2 > 0
        z = torch.sqrt(theta[0]) * torch.randn(N, generator=gen, dtype=theta.dtype)
        return z + torch.randn(N, generator=gen, dtype=theta.dtype), z

    def logLike(x, z, theta):
        return -0.5 * (torch.sum((x - z) ** 2) + torch.sum(z ** 2) / theta[0] + N * torch.log(theta[0]))

    class P(M.TorchMuseProblem):
        def transform_theta(self, theta):
            return np.log(np.asarray(theta, dtype=np.float64))

        def inv_transform_theta(self, theta):
            return np.exp(np.asarray(theta, dtype=np.float64))
    prob = P(None, sample_x_z, logLike)
    res = M.check_self_consistency(prob, [1.7], has_volume_factor=False, atol=1e-5)
    assert max(res.values()) < 1e-5


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_closure_problems_with_their_tensors_on_the_gpu(gpu, M):
    """The same two problems with every tensor on the device (autograd and the solver's vector operations run there), and the
    funnel's estimate next to the HIP engine's on the same data: two implementations of one method, different random streams."""
    res = run_funnel(M, "cuda")
    run_mixing(M, "cuda", nrep=3)
    sample_x_z, logLike = funnel_closures(2000, 1)
    sim = M.TorchMuseProblem(None, sample_x_z, logLike, device="cuda")
    x, _ = sim.sample_x_z(M.SimRng(5, M.DATA_SIM), [0.7])
    hip = M.HipMuseProblem(x.cpu().numpy(), model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, 3.0))
    # get_H! by implicit differentiation: autograd on the device against the HIP engine's closed forms for the same model -- the two
    # draw different random numbers, so the Monte-Carlo means agree to their scatter
    Ht, _ = M.TorchMuseProblem(x, sample_x_z, logLike, device="cuda").implicit_H_batch(3, 0, 6, [0.5], atol=1e-8)
    Hh, _ = hip.implicit_H_batch(3, 0, 64, [0.5], atol=1e-8)
    assert abs(Ht.mean() / Hh.mean() - 1.0) < 0.15
    ref = M.muse(hip, [0.0], rng=11, nsims=400, maxsteps=30, theta_rtol=1e-4, grad_z_logLike_atol=1e-6, alpha=1.0, get_covariance=True)
    sigma = float(np.sqrt(ref.Sigma[0, 0]))
    assert abs(float(res.theta[0]) - float(ref.theta[0])) < 4.0 * sigma * np.sqrt(1 / 60 + 1 / 400)
    hip.close()


def test_a_numpy_subclass_gets_the_interfaces_default_solver(M, O):
    """A subclass of AbstractMuseProblem that states sample_x_z, logLike_and_grad_z_logLike and grad_theta_logLike in numpy -- and no
    ẑ_at_θ -- is solved by the interface's default (src/interface.jl:140-166), as in the reference: the funnel's MAP in closed form,
    the checker's evaluation counts, and a muse() run."""
    N = 500

    class NumpyFunnel(M.AbstractMuseProblem):
        def __init__(self, x):
            self.x = x

        def sample_x_z(self, rng, theta):
            return O.sample_x_z("funnel", N, rng.seed, rng.sim, np.atleast_1d(theta))

        def logLike_and_grad_z_logLike(self, x, z, theta):
            iv = np.exp(-np.atleast_1d(theta)[0])
            return -0.5 * (np.sum((x - z) ** 2) + iv * np.sum(z * z) + N * np.atleast_1d(theta)[0]), (x - z) - iv * z

        def grad_theta_logLike(self, x, z, theta, theta_space=None):
            return np.array([0.5 * (np.exp(-np.atleast_1d(theta)[0]) * np.sum(z * z) - N)])

        def logPrior_theta(self, theta, theta_space=None):
            return -0.5 * float(np.sum(np.atleast_1d(theta) ** 2)) / 9.0

        def grad_logPrior_theta(self, theta, theta_space=None):
            return -np.atleast_1d(theta) / 9.0

        def hess_logPrior_theta(self, theta, theta_space=None):
            return -np.eye(1) / 9.0
    x, _ = O.sample_x_z("funnel", N, 3, M.DATA_SIM, [0.5])
    prob = NumpyFunnel(x)
    zh, info = prob.zhat_at_theta(x, np.zeros(N), [0.5], 1e-8)
    zo, io = O.zhat_at_theta("funnel", x, np.zeros(N), [0.5], 1e-8)
    assert (int(info["iterations"]), int(info["f_calls"]), int(info["status"])) == (io["iterations"], io["f_calls"], io["status"])
    np.testing.assert_allclose(zh, x / (1 + np.exp(-0.5)), atol=1e-9)
    res = M.muse(prob, [0.0], rng=4, nsims=30, maxsteps=20, theta_rtol=1e-3, grad_z_logLike_atol=1e-6, get_covariance=True)
    from test_exact_marginal import exact_scale_family
    mode, sigma = exact_scale_family(x, 1)
    assert abs(res.theta[0] - mode[0]) < 4.0 * sigma[0] / np.sqrt(30) + 0.02 and 0.5 < np.sqrt(res.Sigma[0, 0]) / sigma[0] < 2.0


def test_implicit_differentiation_H_by_autograd(M):
    """get_H!(implicit_diff=true) (src/muse.jl:335-405) for closure problems, every derivative by autograd: against the
    finite-difference branch on the same simulations (tight MAPs), for the funnel -- whose per-simulation H is known in closed form,
    -1/2 e^-θ/(1+e^-θ)^2 ... summed: compared with the finite differences only -- and for the dense mixing model."""
    sample_x_z, logLike = funnel_closures(400, 2)
    prob = M.TorchMuseProblem(None, sample_x_z, logLike)
    theta = np.array([0.3, -0.4])
    Hs, its = prob.implicit_H_batch(9, 0, 3, theta, atol=1e-10)
    assert Hs.shape == (3, 2, 2) and its.shape == (3, 2) and np.all(its >= 1) and np.all(np.abs(Hs[:, 0, 1]) < 1e-10)
    for s in range(3):
        x, z = prob.sample_x_z(M.SimRng(9, s), theta)
        zfid, _ = prob.zhat_at_theta(x, torch.zeros_like(z), theta, 1e-12)
        Hfd = np.empty((2, 2))
        for j in range(2):
            gs = []
            for sign in (+1, -1):
                th = theta.copy()
                th[j] += sign * 1e-4
                xs, _ = prob.sample_x_z(M.SimRng(9, s), th)
                zh, _ = prob.zhat_at_theta(xs, zfid, theta, 1e-12)
                gs.append(prob.grad_theta_logLike(xs, zh, theta))
            Hfd[:, j] = (gs[0] - gs[1]) / 2e-4
        np.testing.assert_allclose(Hs[s], Hfd, rtol=1e-6, atol=1e-6 * np.abs(Hfd).max())
    # through the driver, next to the finite-difference branch; and the dense model
    A, sx, ll = mixing_problem(M, "cpu")
    pm = M.TorchMuseProblem(None, sx, ll)
    x, _ = pm.sample_x_z(M.SimRng(1, M.DATA_SIM), [0.4])
    pm = M.TorchMuseProblem(x, sx, ll, logPrior=gaussian_prior(3.0))
    res = M.muse(pm, [0.0], rng=3, nsims=30, maxsteps=15, theta_rtol=1e-3, grad_z_logLike_atol=1e-6)
    M.get_H_(res, pm, nsims=8, implicit_diff=True)
    Himp = res.H.copy()
    assert len(res.metadata["implicit_diff_cg_hists"]) == 8
    res.Hs, res.H = [], None
    M.get_H_(res, pm, nsims=8, grad_z_logLike_atol=1e-8)
    np.testing.assert_allclose(Himp, res.H, rtol=0.03)


def test_solver_against_the_checker_on_random_problems(O):
    """optim.lbfgs and the checker's zhat_at_theta on the SAME objective (the checker's own logLike_and_grad_z: identical floats, so
    every decision of the line search must fall the same way): random models / sizes / parameters / tolerances / starts -- equal
    iteration and evaluation counts, status, and the same MAP."""
    from museinference_jl_amd import optim
    rng = np.random.default_rng(3)
    longest = 0
    for case in range(80):
        model = str(rng.choice(["funnel", "noise", "smooth"]))
        N = int(rng.integers(8, 1200))
        nth = 1 if model == "noise" else min(int(rng.integers(1, 5)), N)
        theta = rng.uniform(-2, 2, size=nth)
        atol = float(rng.choice([1e-2, 1e-5, 1e-8]))
        x, z = O.sample_x_z(model, N, int(rng.integers(1, 1 << 40)), int(rng.integers(0, 100)), theta)
        z0 = np.zeros(N) if rng.random() < 0.5 else z

        def fg(zz):
            f, g = O.logLike_and_grad_z(model, x, zz.numpy(), theta)
            return -f, -torch.as_tensor(g)
        zh, info = optim.lbfgs(fg, torch.as_tensor(z0), atol)
        zo, io = O.zhat_at_theta(model, x, z0, theta, atol)
        assert (info["iterations"], info["f_calls"], info["status"]) == (io["iterations"], io["f_calls"], io["status"]), (case, model, N, info, io)
        np.testing.assert_allclose(zh.numpy(), zo, rtol=0, atol=1e-9)
        longest = max(longest, io["iterations"])
    assert longest >= 20        # (the stencil model's solves: real line searches and a wrapping history)
