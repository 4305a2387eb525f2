"""-m gpu: the SURVEY.md §8 rows beyond the plain map, on the HIP path -- BASELINE.json's configs at their own
sizes (configs[2] at nsims = 128, configs[3]'s finite-difference H at 64 sims, configs[4]'s N = 10^5 / 8 theta),
the muse! keyword variants (row f2), checkpoint / resume / save_MAPs (row f3), theta transforms (row f4), the
element split of strongly scaled maps (row e2) and cluster placement under the gathered (RCCL) map."""
import os

import numpy as np
import pytest

from test_gpu_parity import assert_same_path_or_close
from test_host import LogNormalVariancePrior, check_outer_variants

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


# ---- BASELINE.json configs at their own sizes ------------------------------------------------------
def test_config5_smooth_1e5_8theta_vs_oracle(gpu, M, O):
    """configs[4]: hierarchical linear-Gaussian (stencil) model, N = 10^5, 8 theta blocks -- cluster placement.
    Two sims against the oracle.  Iteration-count parity is empirical, not a theorem (tree vs
    sequential sums can move an evaluation count on long solves); where the counts agree the scores must agree to
    rtol 1e-10, and in any case to the accuracy the MAP tolerance implies."""
    N, nth, atol = 100000, 8, 1e-2
    th = [1.0, 2.0, 3.0, 0.5, 0.0, -1.0, 1.5, 2.5]
    prob = M.HipMuseProblem(None, model="smooth", ntheta=nth, N=N)
    g, info = prob.map_and_score_batch(21, 0, 2, th, atol=atol, z0_mode=0)
    go, zo, io = O.map_and_score_batch("smooth", N, 21, 0, 2, th, atol=atol, z0_mode=0, nthreads=2)
    assert np.all(info["status"] == 0) and np.all(io["status"] == 0)
    zh = prob.get_zhat(0, 2)
    for k in range(2):
        if (info["iterations"][k], info["f_calls"][k]) == (io["iterations"][k], io["f_calls"][k]):
            np.testing.assert_allclose(g[k], go[k], rtol=1e-10)
            np.testing.assert_allclose(zh[k], zo[k], rtol=0, atol=1e-9)
        else:  # both converged to ||grad||_inf <= atol on different paths: |dz| <= 2 atol / lambda_min(Hessian)
            lam = np.exp(-max(th))
            assert np.abs(zh[k] - zo[k]).max() <= 2 * atol / lam
            np.testing.assert_allclose(g[k], go[k], rtol=0, atol=2 * atol / lam * np.sqrt(N))
    prob.close()


def test_config5_smooth_1e5_full_cluster_load(gpu, M):
    """configs[4] per-GPU share (128 elements, every CU carrying two workgroups of a cluster): all converged, no
    cluster-timeout flag (a MuseError would be raised), two launches bitwise equal."""
    N, nth = 100000, 8
    th = [1.0] * nth
    prob = M.HipMuseProblem(None, model="smooth", ntheta=nth, N=N)
    g1, i1 = prob.map_and_score_batch(0, 0, 128, th, atol=1e-2, z0_mode=0)
    g2, i2 = prob.map_and_score_batch(0, 0, 128, th, atol=1e-2, z0_mode=0)
    assert np.all(i1["status"] == 0)
    assert np.array_equal(g1, g2) and np.array_equal(i1, i2)
    assert i1["iterations"].min() >= 3
    prob.close()


def test_config3_noise_1e6_nsims128(gpu, M):
    """configs[2] at its own size: N = 10^6, nsims = 128 (128 clusters of 4 workgroups = the full 512-workgroup
    grid): every solve converged, closed-form MAP and score on three sims, score moments over the batch."""
    N, S, theta = 1000000, 128, 0.5
    prob = M.HipMuseProblem(None, model="noise", ntheta=1, N=N)
    g, info = prob.map_and_score_batch(3, 0, S, [theta], atol=1e-2, z0_mode=0)
    assert np.all(info["status"] == 0)
    e = np.exp(theta)
    for k in (0, 64, 127):
        x, _ = prob.sample_x_z(M.SimRng(3, k), [theta])
        zh = prob.get_zhat(k, k + 1)[0]
        np.testing.assert_allclose(zh, x / (1 + e), rtol=0, atol=1e-9)
        np.testing.assert_allclose(g[k, 0], 0.5 * (np.exp(-theta) * np.sum((x - zh) ** 2) - N), rtol=1e-11)
    # E[s] = -N / (2 (1 + e^theta)),  Var[s] = N e^{2 theta} / (2 (1 + e^theta)^2)  (same algebra)
    mean, var = -N / (2 * (1 + e)), N * e**2 / (2 * (1 + e) ** 2)
    assert abs(g.mean() - mean) < 4 * np.sqrt(var / S)
    prob.close()


def test_config4_fd_jacobian_64_sims(gpu, M, O):
    """configs[3]: 4-block funnel, N = 10^4, finite-difference H at 64 sims (512 perturbed MAPs + normals-only
    elements in the fiducial launch): spot checks against the oracle, the closed-form CRN Jacobian on the diagonal
    blocks, H = mean(Hs) against J (Gaussian model: H = J up to Monte-Carlo error)."""
    N, nth, S = 10000, 4, 64
    th = np.array([1.0, 0.5, -0.5, 2.0])
    step = np.full(nth, 0.05)
    prob = M.HipMuseProblem(None, model="funnel", ntheta=nth, N=N)
    Hs, info = prob.fd_jacobian_batch(11, 0, S, th, step, atol=1e-2, fid_mode=0)
    assert np.all(info["status"] == 0)
    _, zfid, _ = O.map_and_score_batch("funnel", N, 11, M.MASTER_SIM, M.MASTER_SIM + 1, th, atol=1e-2, z0_mode=0)
    for s in (0, 31, 63):
        Ho = O.fd_jacobian("funnel", N, 11, s, th, step, zfid[0], atol=1e-2)
        np.testing.assert_allclose(Hs[s], Ho, rtol=1e-8, atol=1e-8 * np.abs(Ho).max())
    H = Hs.mean(axis=0)
    Jexact = 0.5 * (N / nth) * np.exp(2 * th) / (1 + np.exp(th)) ** 2          # per block: N_k e^{2θ}/(2(1+e^θ)²)
    assert np.all(np.abs(np.diag(H) / Jexact - 1) < 5 / np.sqrt(S * N / nth) + 2e-3)   # FD truncation O(step²)
    off = H - np.diag(np.diag(H))
    assert np.abs(off).max() < 1e-6 * Jexact.max()                            # blocks are independent
    prob.close()


@pytest.mark.parametrize("N,nth", [(10000, 4), (10000, 1), (7001, 3), (4097, 8)])
def test_fd_launch_that_carries_its_fiducial_changes_no_bit(gpu, M, O, N, nth):
    """get_H!'s finite-difference map (src/muse.jl:417-442) once the simulations' normals are cached, as ONE launch whose problem 0 is the
    fiducial MAP and whose other problems draw their x and then wait for its tag (round 6: built, 8 us slower per call than the two
    launches, left off; debug flag bit 20 switches it on) -- against the two launches, against the first call of a fresh context
    (nothing cached), in whole Jacobians and in column ranges that begin and end inside a simulation's Jacobian; and against the oracle."""
    th = np.linspace(0.4, 1.3, nth)
    step = np.full(nth, 0.05)
    S = 70
    prob = M.HipMuseProblem(None, model="funnel", ntheta=nth, N=N)
    first, i1 = prob.fd_jacobian_batch(21, 3, 3 + S, th, step, atol=1e-2)         # fills the cache: fiducial launch + perturbed launch
    two, i3 = prob.fd_jacobian_batch(21, 3, 3 + S, th, step, atol=1e-2)           # cache held, two launches (the default): the fiducial's
                                                                                  # normals drawn by a kernel of their own (round 6) ...
    prob.debug_flags(1 << 21)
    own, i4 = prob.fd_jacobian_batch(21, 3, 3 + S, th, step, atol=1e-2)           # ... against the fiducial problem drawing them itself
    prob.debug_flags(0)
    assert np.array_equal(own, two) and i4.tobytes() == i3.tobytes()
    prob.debug_flags(1 << 20)
    folded, i2 = prob.fd_jacobian_batch(21, 3, 3 + S, th, step, atol=1e-2)        # cache held: the one launch
    assert np.array_equal(first, folded) and np.array_equal(two, folded)
    assert i1.tobytes() == i2.tobytes() == i3.tobytes() and np.all(i2["status"] == 0)
    lo, hi = 2 * nth + (1 if nth > 1 else 0), 41 * nth - (1 if nth > 1 else 0)   # units of the list (sim 3, column 0), (sim 3, column 1), ...
    cols, ic = prob.fd_jacobian_columns(21, 3, lo, hi, th, step, atol=1e-2)
    whole = folded.transpose(0, 2, 1).reshape(-1, nth)                            # unit e = (sim, column j): d g_i / d theta_j over i
    assert np.array_equal(cols, whole[lo:hi])
    # a different theta on the same context: the flag's tag moves on, nothing of the previous call's fiducial is taken
    th2 = th + 0.3
    a, _ = prob.fd_jacobian_batch(21, 3, 3 + S, th2, step, atol=1e-2)
    prob.debug_flags(0)
    b, _ = prob.fd_jacobian_batch(21, 3, 3 + S, th2, step, atol=1e-2)
    assert np.array_equal(a, b) and not np.array_equal(a, folded)
    _, zfid, _ = O.map_and_score_batch("funnel", N, 21, M.MASTER_SIM, M.MASTER_SIM + 1, th, atol=1e-2, z0_mode=0)
    for s in (0, S - 1):
        Ho = O.fd_jacobian("funnel", N, 21, 3 + s, th, step, zfid[0], atol=1e-2)
        np.testing.assert_allclose(folded[s], Ho, rtol=1e-8, atol=1e-8 * np.abs(Ho).max())
    prob.close()


# ---- row f2: muse! keyword variants on the HIP path ---------------------------------------------------
def test_outer_loop_variants_on_hip(gpu, M):
    """Broyden / diagonal-Broyden (with memory limit, with a user H^-1_like'), callable alpha + regularize, and
    the transformed-theta front-end, each a full muse! run on HipMuseProblem against the golden trajectories of the
    independent restatement (tests/muse_reference.py on the oracle's map).  rtol 1e-8: the scores agree to 1e-10
    and the Newton iteration propagates them."""
    def make(x, prior="gauss"):
        return M.HipMuseProblem(x, model="funnel", ntheta=4, prior=M.GaussianPrior(0.0, 3.0) if prior == "gauss" else None)
    check_outer_variants(M, make, rtol=1e-8)


def test_native_muse_run_matches_independent_restatement(gpu, M):
    """muse_run (the outer loop in the library's C host code) against the golden trajectory whose algebra comes
    from tests/muse_reference.py -- Newton step, diagonal H^-1 = -1/var, H^-1_post', convergence test."""
    d = np.load(os.path.join(HERE, "golden", "muse_trajectory.npz"))
    prob = M.HipMuseProblem(d["x"], model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, 3.0))
    for native in (True, False):
        res = M.muse(prob, [1.0], rng=42, nsims=32, get_covariance=True, native=native)
        np.testing.assert_allclose(np.array([h["θ"] for h in res.history]), d["thetas"], rtol=1e-9)
        np.testing.assert_allclose(np.array([h["g_like′"] for h in res.history]), d["g_like"], rtol=1e-9)
        np.testing.assert_allclose(np.array([h["H⁻¹_post′"] for h in res.history]), d["Hinv_post"], rtol=1e-9)
        np.testing.assert_allclose(res.theta, d["theta"], rtol=1e-8)
        np.testing.assert_allclose(res.J, d["J"], rtol=1e-9)
        np.testing.assert_allclose(np.array(res.Hs), d["Hs"], rtol=1e-7)
        np.testing.assert_allclose(res.Sigma, d["Sigma"], rtol=1e-7)
    d4 = np.load(os.path.join(HERE, "golden", "muse_outer_variants.npz"))
    p4 = M.HipMuseProblem(d4["x"], model="funnel", ntheta=4, prior=M.GaussianPrior(0.0, 3.0))
    res = M.muse(p4, [1.0] * 4, rng=1, nsims=24, maxsteps=7, theta_rtol=0.0, native=True)
    np.testing.assert_allclose(np.array([h["θ"] for h in res.history]), d4["sims_thetas"][:-1], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(res.theta, d4["sims_thetas"][-1], rtol=1e-8, atol=1e-8)
    # (muse_run's DomainError exit cannot be provoked through its own option set: -1/var < 0 and a flat or Gaussian
    #  prior keep H^-1_post' negative definite; the Python driver's exit is tested on CPU with a user H^-1_like'.)
    p4.close()
    prob.close()


@pytest.mark.parametrize("model,N,nth,nsims,maxsteps,rtol", [
    ("funnel", 10000, 1, 512, 50, 1e-1), ("funnel", 10000, 1, 64, 6, 0.0), ("funnel", 3000, 4, 100, 50, 1e-1),
    ("noise", 5000, 1, 33, 50, 1e-1), ("smooth", 2000, 3, 17, 9, 1e-2), ("funnel", 512, 8, 40, 12, 1e-3)])
def test_device_resident_muse_loop_equals_host_loop(gpu, M, O, model, N, nth, nsims, maxsteps, rtol):
    """muse_run_device -- the per-iteration algebra (means, corrected variances, prior terms, H^-1_post', Newton step,
    convergence test: src/muse.jl:163-166,177-232) in a step kernel on the GPU, the next theta read by the next map from
    device memory, no host round trip between two maps -- against muse_run, the same loop with the algebra on the host:
    the same number of iterations and the same bits in every history record, score, solver info and in the result."""
    xdata, _ = O.sample_x_z(model, N, 3, M.DATA_SIM, np.zeros(nth))
    prior = M.GaussianPrior(0.0, 3.0) if nth != 4 else None
    prob = M.HipMuseProblem(xdata, model=model, ntheta=nth, prior=prior)
    th0 = np.linspace(1.0, 0.3, nth)
    outs = []
    for dev in (False, True, True):
        n, theta, hist, gs, info = prob.run_muse(11, th0, nsims=nsims, maxsteps=maxsteps, theta_rtol=rtol, atol=1e-2, alpha=0.7,
                                                 device_loop=dev)
        outs.append((n, theta, hist, gs, info, prob.get_zhat(0, nsims + 1)))
    n0, theta0, hist0, gs0, info0, zh0 = outs[0]
    assert 2 <= n0 <= maxsteps
    for n, theta, hist, gs, info, zh in outs[1:]:
        assert n == n0
        # the MAPs the loop leaves in memory -- a worker's last one is carried in registers from iteration to iteration and
        # stored when the loop ENDS (round 6), on convergence as at maxsteps
        assert np.array_equal(zh, zh0)
        assert np.array_equal(theta, theta0)
        assert np.array_equal(hist[:, :-1], hist0[:, :-1])       # (last column: the iteration's wall time)
        assert np.all(hist[:, -1] > 0) and np.all(hist[:, -1] < 1.0)
        assert np.array_equal(gs, gs0) and np.array_equal(info, info0)
    # and through the driver (which takes the host loop)
    res = M.muse(prob, th0, rng=11, nsims=nsims, maxsteps=maxsteps, theta_rtol=rtol, alpha=0.7)
    assert len(res.history) == n0 and np.array_equal(res.theta, theta0)
    # a z0 warm start (first map from the resident MAPs) and the host's knowledge lagging behind the device's stop
    n1, t1, h1, g1, i1 = prob.run_muse(11, th0, nsims=nsims, maxsteps=maxsteps, theta_rtol=rtol, atol=1e-2, alpha=0.7, z0_warm=True,
                                       device_loop=False)
    prob.run_muse(11, th0, nsims=nsims, maxsteps=2, theta_rtol=rtol, atol=1e-2, alpha=0.7, device_loop=True)   # leaves other MAPs behind
    prob.run_muse(11, th0, nsims=nsims, maxsteps=maxsteps, theta_rtol=rtol, atol=1e-2, alpha=0.7, device_loop=False)
    n2, t2, h2, g2, i2 = prob.run_muse(11, th0, nsims=nsims, maxsteps=maxsteps, theta_rtol=rtol, atol=1e-2, alpha=0.7, z0_warm=True,
                                       device_loop=True)
    assert n2 == n1 and np.array_equal(t2, t1) and np.array_equal(h2[:, :-1], h1[:, :-1]) and np.array_equal(i2, i1)
    # a map after the loop is unaffected by the loop's device-side theta and stop flag
    g, _ = prob.map_and_score_batch(11, 0, 5, th0)
    ref = M.HipMuseProblem(xdata, model=model, ntheta=nth, prior=prior)
    gr, _ = ref.map_and_score_batch(11, 0, 5, th0)
    assert np.array_equal(g, gr)
    prob.close()
    ref.close()


@pytest.mark.parametrize("model,N,nth,nsims", [
    ("funnel", 10000, 1, 512), ("funnel", 10000, 1, 255), ("funnel", 10000, 1, 256), ("funnel", 9999, 1, 700), ("noise", 10000, 1, 515),
    ("funnel", 10000, 3, 530), ("funnel", 3000, 4, 1100), ("funnel", 512, 2, 1300)])
def test_loop_kernel_layouts_change_no_bit(gpu, M, O, model, N, nth, nsims):
    """Round 5: with more elements than workgroups the loop kernel's stepper solves elements too -- the last in the deal: it has the
    fewest -- and sweeps the scores when its own are done.  Which workgroup solves an element changes no bit: the layouts
    (muse_debug_flags bit 6: the data element is the stepper's own; bit 7: a stepper that only steps, the layout before) and the host
    loop return the same history, scores and solver records."""
    xdata, _ = O.sample_x_z(model, N, 3, M.DATA_SIM, np.zeros(nth))
    prob = M.HipMuseProblem(xdata, model=model, ntheta=nth, prior=M.GaussianPrior(0.0, 3.0))
    th0 = np.linspace(0.8, 0.2, nth)
    kw = dict(nsims=nsims, maxsteps=5, theta_rtol=0.0, atol=1e-2, alpha=0.7)
    want = prob.run_muse(4, th0, device_loop=False, **kw)
    zh = prob.get_zhat(0, nsims + 1)
    for flags in (0, 64, 128, 512, 0):     # (512: every MAP stored in every iteration, as before round 6)
        assert prob._lib.muse_debug_flags(prob._ctx, flags) == 0
        got = prob.run_muse(4, th0, device_loop=True, **kw)
        assert np.array_equal(prob.get_zhat(0, nsims + 1), zh), flags
        assert got[0] == want[0] == 5 and np.array_equal(got[1], want[1]) and np.array_equal(got[2][:, :-1], want[2][:, :-1])
        assert np.array_equal(got[3], want[3]) and np.array_equal(got[4], want[4]), flags
    assert prob._lib.muse_debug_flags(prob._ctx, 0) == 0
    prob.close()
    # ... and a sharded loop's rank with more elements than workgroups (one rank in the communicator: its stepper reads the board)
    if nsims in (512, 700):
        shp = M.HipMuseProblem(xdata, model=model, ntheta=nth, prior=M.GaussianPrior(0.0, 3.0))
        shp.comm_init(1, 0, M.HipMuseProblem.comm_unique_id("shm", (nsims + 1) * nth))
        for flags in (0, 64, 128):
            assert shp._lib.muse_debug_flags(shp._ctx, flags) == 0
            got = shp.run_muse_sharded(4, th0, **kw)
            assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2][:, :-1], want[2][:, :-1])
            assert np.array_equal(got[3], want[3]), flags
        shp.close()


def test_device_loop_reports_errors_like_the_host_loop(gpu, M):
    """A step that cannot be taken ends both loops with the same error: a NaN theta makes every score NaN, the score
    variance NaN and H^-1_like' singular (the device loop's step kernel raises its stop flag, the launches already
    enqueued drain as no-ops)."""
    prob = M.HipMuseProblem(np.linspace(-1, 1, 600), model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, 3.0))
    for dev in (False, True):
        with pytest.raises(M.MuseError, match="singular"):
            prob.run_muse(1, [np.nan], nsims=8, maxsteps=5, theta_rtol=0.1, atol=1e-2, alpha=0.7, device_loop=dev)
    g, _ = prob.map_and_score_batch(1, 0, 4, [0.5])              # the context is usable afterwards
    assert np.all(np.isfinite(g))
    n, theta, hist, gs, info = prob.run_muse(1, [0.5], nsims=8, maxsteps=5, theta_rtol=0.1, atol=1e-2, alpha=0.7)
    assert n >= 2 and np.all(np.isfinite(theta))
    prob.close()


def test_loop_kernel_that_cannot_stay_resident_falls_back_to_the_host_loop(gpu, M, monkeypatch):
    """The loop kernel's workgroups meet once per iteration, so all of them must be resident.  When they cannot be (here: a
    test hook launches one worker per element, 301 workgroups of a placement that fits 256), its bounded waits expire, the
    call reports it -- and run_muse, left to choose the loop itself, runs the host loop instead: the same bits as an
    undisturbed run.  The context stays usable."""
    x = np.cos(0.1 * np.arange(10000))
    prob = M.HipMuseProblem(x, model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, 3.0))
    kw = dict(nsims=300, maxsteps=3, theta_rtol=0.0, atol=1e-2, alpha=0.7)
    want = prob.run_muse(5, [0.8], device_loop=False, **kw)
    prob.debug_flags(M.HipMuseProblem.DEBUG_LOOP_OVERSUBSCRIBE)   # (the environment switch of the same name is read at context creation)
    with pytest.raises(M.MuseError, match="not all resident"):
        prob.run_muse(5, [0.8], device_loop=True, **kw)
    with pytest.warns(RuntimeWarning, match="host loop"):
        got = prob.run_muse(5, [0.8], **kw)
    prob.debug_flags(0)
    assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2][:, :-1], want[2][:, :-1])
    assert np.array_equal(got[3], want[3])
    again = prob.run_muse(5, [0.8], device_loop=True, **kw)     # the loop kernel itself, afterwards
    assert again[0] == want[0] and np.array_equal(again[1], want[1]) and np.array_equal(again[3], want[3])
    prob.close()


def test_aborted_loop_leaves_no_normals_cache_tag_behind(gpu, M):
    """A device-resident loop whose FIRST iteration stores the simulations' standard normals (the cross-call normals cache) and
    that dies before all of them are written -- a test hook makes every second worker leave before its first solve; the stepper's
    bounded wait then expires -- must not leave the cache tagged as holding the range: the maps that follow over the same
    simulations (the second stores, the third loads) equal those of a fresh context bit for bit.  (Round 4 claimed the tag when
    the launch was ENQUEUED: the next map loaded slots that had never been written.)"""
    x = np.sin(0.05 * np.arange(10000))
    prob = M.HipMuseProblem(x, model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, 3.0))
    kw = dict(nsims=40, maxsteps=3, theta_rtol=0.0, atol=1e-2, alpha=0.7)
    assert prob._lib.muse_debug_flags(prob._ctx, 16) == 0
    with pytest.raises(M.MuseError):
        prob.run_muse(9, [0.7], device_loop=True, **kw)
    assert prob._lib.muse_debug_flags(prob._ctx, 0) == 0
    fresh = M.HipMuseProblem(x, model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, 3.0))
    fresh.set_normals_cache(False)
    want, winfo = fresh.map_and_score_batch(9, 0, 40, [0.7], include_data=True, atol=1e-2, z0_mode=M.Z0_ZERO)
    for _ in range(4):   # (plain maps: seen, stored, loaded, loaded)
        got, info = prob.map_and_score_batch(9, 0, 40, [0.7], include_data=True, atol=1e-2, z0_mode=M.Z0_ZERO)
        assert np.array_equal(got, want) and np.array_equal(info, winfo)
    # ... and the native loops afterwards are the undisturbed ones (the failed call is remembered: the host loop runs)
    a = prob.run_muse(9, [0.7], **kw)
    b = fresh.run_muse(9, [0.7], device_loop=False, **kw)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
    prob.close()
    fresh.close()


# ---- row f3: checkpoint / resume / save_MAPs on the HIP path ---------------------------------------------
def test_checkpoint_resume_and_saved_maps_on_hip(gpu, M, O, tmp_path):
    N = 3000
    x, _ = O.sample_x_z("funnel", N, 5, M.DATA_SIM, [0.0])
    kw = dict(rng=11, nsims=16, theta_rtol=0.0, grad_z_logLike_atol=1e-9)
    prob = M.HipMuseProblem(x, model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, 3.0))
    full = M.muse(prob, [1.0], maxsteps=4, save_MAPs=True, **kw)
    h = full.history[-1]
    # history's saved MAPs are the resident slots at that iteration: the last record equals what the context holds
    assert np.array_equal(h["ẑ_dat"], prob.get_zhat(0, 1)[0])
    assert np.array_equal(np.array(h["ẑ_sims"]), prob.get_zhat(1, 17))
    np.testing.assert_allclose(h["ẑ_dat"], x / (1 + np.exp(-h["θ"][0])), rtol=0, atol=1e-8)   # closed-form MAP
    # a preprocessing hook (src/muse.jl:102-104: e.g. device -> host conversion) is applied to every saved MAP
    r = M.muse(prob, [1.0], maxsteps=1, save_MAPs=lambda z: float(np.sum(z)), **kw)
    assert isinstance(r.history[0]["ẑ_dat"], float) and len(r.history[0]["ẑ_sims"]) == 16
    # checkpoint after every iteration (src/muse.jl:234), resume from the file in a NEW problem object
    ck = str(tmp_path / "ck.pkl")
    M.muse(prob, [1.0], maxsteps=2, checkpoint_filename=ck, **kw)
    prob.close()
    res = M.load_result(ck)
    assert len(res.history) == 2 and res.rng == 11
    prob2 = M.HipMuseProblem(x, model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, 3.0))
    M.muse_(res, prob2, maxsteps=4, nsims=16, theta_rtol=0.0, grad_z_logLike_atol=1e-9)
    assert len(res.history) == 4
    # a resumed run restarts its MAPs from zero(z) (src/muse.jl:151), so the iterates agree to the MAP tolerance
    np.testing.assert_allclose(res.theta, full.theta, rtol=0, atol=1e-7)
    np.testing.assert_allclose(res.history[2]["θ"], full.history[2]["θ"], rtol=1e-12)   # θ of record 3 = step of record 2
    M.get_J_(res, prob2, nsims=16)
    M.get_H_(res, prob2, nsims=2)
    assert res.Sigma is not None and res.dist.σ > 0
    prob2.close()


# ---- row f4: theta transforms on the HIP path --------------------------------------------------------------
def test_positive_theta_front_end_on_hip(gpu, M, O):
    N, nth = 2000, 2
    x, _ = O.sample_x_z("funnel", N, 9, M.DATA_SIM, [0.0] * nth)
    base = M.HipMuseProblem(x, model="funnel", ntheta=nth)
    prob = M.PositiveThetaProblem(base, prior=LogNormalVariancePrior())
    res = M.check_self_consistency(prob, [1.7, 0.6], atol=1e-2)       # src/interface.jl:209-230, reference atol
    assert max(res.values()) < 1e-4
    # scores_in_both_spaces through the batched HIP seam against the per-simulation operators
    v = np.array([1.7, 0.6])
    g_eng, info = prob.map_and_score_batch(4, 0, 5, v, include_data=True, atol=1e-8)
    g_u, g_t = prob.scores_in_both_spaces(g_eng, v, np.log(v))
    zh = base.get_zhat(0, 6)
    for e in range(6):
        xe = x if e == 0 else base.sample_x_z(M.SimRng(4, e - 1), np.log(v))[0]
        np.testing.assert_allclose(g_u[e], prob.grad_theta_logLike(xe, zh[e], v, M.UnTransformedθ), rtol=1e-10)
        np.testing.assert_allclose(g_t[e], prob.grad_theta_logLike(xe, zh[e], np.log(v), M.Transformedθ), rtol=1e-10)
    # a full run in the transformed space lands where the log-variance run does
    r_t = M.muse(prob, [np.e, np.e], rng=4, nsims=40, get_covariance=True)
    r_u = M.muse(M.HipMuseProblem(x, model="funnel", ntheta=nth, prior=M.GaussianPrior(0.0, 3.0)), [1.0, 1.0], rng=4,
                 nsims=40, get_covariance=True)
    np.testing.assert_allclose(np.log(r_t.theta), r_u.theta, atol=5e-3)
    v_hat = r_t.theta
    np.testing.assert_allclose(r_t.H * np.outer(v_hat, v_hat), r_u.H, rtol=2e-2, atol=2e-2 * np.abs(r_u.H).max())
    # implicit-diff H in the variance space = engine H / (v vᵀ)  (no silent pass-through of variances)
    Hs_v, _ = prob.implicit_H_batch(4, 0, 3, v)
    Hs_t, _ = base.implicit_H_batch(4, 0, 3, np.log(v))
    np.testing.assert_allclose(Hs_v, Hs_t / np.outer(v, v), rtol=1e-14)
    base.close()


# ---- row e2: one element split over several workgroups (strongly scaled maps) -------------------------------
@pytest.mark.parametrize("model,N,nth,theta", [
    ("funnel", 10000, 1, [1.0]), ("funnel", 10000, 4, [1.0, 0.5, -0.5, 2.0]), ("noise", 8191, 1, [0.3]),
    ("smooth", 6000, 3, [1.0, 2.0, 0.5]), ("funnel", 777, 1, [-0.5])])
@pytest.mark.parametrize("split", [2, 4, 8])
def test_element_split_vs_oracle_and_invariances(gpu, M, O, model, N, nth, theta, split):
    """64 elements on 256 compute units: `split` workgroups per element.  Against the oracle (identical counts,
    scores rtol 1e-10, MAPs 1e-9); bitwise run to run; bitwise independent of the batch (a sim's result does not
    depend on how many elements share the launch -- what makes an N-GPU run equal the 1-GPU run at the same split);
    and within 1e-12 relative of the unsplit result (the split only changes the summation tree)."""
    xdata, _ = O.sample_x_z(model, N, 77, M.DATA_SIM, np.zeros(nth))
    prob = M.HipMuseProblem(xdata, model=model, ntheta=nth)
    g0, i0 = prob.map_and_score_batch(42, 0, 24, theta, include_data=True, atol=1e-2)
    prob.set_element_split(split)
    g, info = prob.map_and_score_batch(42, 0, 24, theta, include_data=True, atol=1e-2)
    zh = prob.get_zhat(0, 25)
    go, zo, io = O.map_and_score_batch(model, N, 42, 0, 24, theta, atol=1e-2, x_data=xdata, z0_mode=0)
    # elements on the oracle's L-BFGS path: scores rtol 1e-10, MAPs 1e-9; an element that left it (the split changes the
    # summation tree; long stencil solves only) is still BOUNDED: both sides converged, |dz| <= 2 atol / lambda_min
    same = assert_same_path_or_close(info, io, zh, zo, g, go, 1e-2, theta, model, f"split {split}")
    assert same.mean() >= 0.9
    g2, info2 = prob.map_and_score_batch(42, 0, 24, theta, include_data=True, atol=1e-2)
    assert np.array_equal(g, g2) and np.array_equal(info, info2) and np.array_equal(zh, prob.get_zhat(0, 25))
    same0 = (info["iterations"] == i0["iterations"]) & (info["f_calls"] == i0["f_calls"])
    np.testing.assert_allclose(g[same0], g0[same0], rtol=1e-12)
    # warm restart at the MAPs: no iteration, identical scores
    g3, i3 = prob.map_and_score_batch(42, 0, 24, theta, include_data=True, atol=1e-2, z0_mode=M.Z0_WARM)
    assert np.all(i3["iterations"] == 0) and np.array_equal(g3, g)
    gp, ip = prob.map_and_score_batch(42, 10, 17, theta, atol=1e-2)      # (overwrites the resident MAPs of slots 0..6)
    assert np.array_equal(gp, g[11:18]) and np.array_equal(ip, info[11:18])
    prob.set_element_split(0)
    g4, _ = prob.map_and_score_batch(42, 0, 24, theta, include_data=True, atol=1e-2)
    assert np.array_equal(g4, g0)
    prob.close()


def test_element_split_drivers_and_errors(gpu, M, O):
    """A whole muse! + get_J! + get_H! (FD and implicit) run with split elements: same estimates as the unsplit run to
    the accuracy the split's summation order allows; bad splits are rejected."""
    x, _ = O.sample_x_z("funnel", 10000, 5, M.DATA_SIM, [0.0, 0.0])
    a = M.HipMuseProblem(x, model="funnel", ntheta=2, prior=M.GaussianPrior(0.0, 3.0))
    b = M.HipMuseProblem(x, model="funnel", ntheta=2, prior=M.GaussianPrior(0.0, 3.0))
    b.set_element_split(4)
    ra = M.muse(a, [1.0, 0.5], rng=3, nsims=31, maxsteps=4, get_covariance=True)
    rb = M.muse(b, [1.0, 0.5], rng=3, nsims=31, maxsteps=4, get_covariance=True)
    np.testing.assert_allclose(rb.theta, ra.theta, rtol=1e-9)
    np.testing.assert_allclose(rb.J, ra.J, rtol=1e-9)
    np.testing.assert_allclose(rb.H, ra.H, rtol=1e-6, atol=1e-6 * np.abs(ra.H).max())
    Ha, _ = a.implicit_H_batch(3, 0, 3, ra.theta)
    Hb, _ = b.implicit_H_batch(3, 0, 3, ra.theta)
    np.testing.assert_allclose(Hb, Ha, rtol=1e-8, atol=1e-8 * np.abs(Ha).max())
    with pytest.raises(M.MuseError):
        b.set_element_split(3)
    with pytest.raises(M.MuseError):
        b.set_element_split(32)
    a.close()
    b.close()


# ---- cluster placement under the gathered (RCCL) map -------------------------------------------------------
@pytest.mark.parametrize("model,N,nth,theta,nel,split", [
    ("smooth", 66001, 2, [1.0, 2.5], 128, 0),      # cluster placement by N
    ("funnel", 10000, 1, [1.0], 64, 4)])           # cluster placement by split (the strongly scaled bench shape)
def test_cluster_placement_under_gathered_map(gpu, M, model, N, nth, theta, nel, split):
    """muse_map_and_score_batch_gather_async with workgroup clusters: the RCCL all-gather of step k runs on the
    high-priority stream while the cluster solver of step k+1 is resident and spinning on its counters.  One rank,
    pipelined over the four result areas: no cluster-timeout flag, bitwise equal to the plain map."""
    p = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
    if split:
        p.set_element_split(split)
    p.comm_init(1, 0, M.HipMuseProblem.comm_unique_id())
    ref = [p.map_and_score_batch(0, b, b + nel, theta) for b in (0, nel, 2 * nel)]
    pend, out = [], []
    for k in range(9):
        b = (k % 3) * nel
        n = p.map_and_score_batch_gather_async(0, b, b + nel, theta, nel, result_area=k % 4)
        pend.append((n, k % 4))
        if len(pend) > 3:
            out.append(p.batch_wait_gathered(nel, nel, pend.pop(0)[1]))
    while pend:
        out.append(p.batch_wait_gathered(nel, nel, pend.pop(0)[1]))
    for k, (g_all, info) in enumerate(out):
        assert np.array_equal(g_all[0], ref[k % 3][0]) and np.array_equal(info, ref[k % 3][1])
        assert np.all(info["status"] == 0)
    p.close()


# ---- get_H! over Jacobian columns (the reference's other parallel axis, src/muse.jl:327-333) ----------------
@pytest.mark.parametrize("model,N,nth,theta", [
    ("funnel", 10000, 4, [1.0, 0.5, -0.5, 2.0]), ("smooth", 2000, 3, [1.0, 2.0, 0.5]), ("funnel", 70001, 2, [0.3, 0.1])])
@pytest.mark.parametrize("fid_mode", [0, 1])
def test_H_column_ranges_equal_whole_jacobians(gpu, M, model, N, nth, theta, fid_mode):
    """A column range that begins and ends inside a simulation's Jacobian gives exactly the columns of the
    whole-simulation call, for the finite-difference and the implicit-differentiation branch."""
    prob = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
    step = np.full(nth, 0.05)
    nsims, s0 = 3, 2
    Hs, info = prob.fd_jacobian_batch(11, s0, s0 + nsims, theta, step, atol=1e-2, fid_mode=fid_mode)
    allcols = Hs.transpose(0, 2, 1).reshape(nsims * nth, nth)
    for lo, hi in ((0, nsims * nth), (1, nth + 1), (nth - 1, 2 * nth + 1), (nsims * nth - 1, nsims * nth)):
        cols, ci = prob.fd_jacobian_columns(11, s0, lo, hi, theta, step, atol=1e-2, fid_mode=fid_mode)
        assert np.array_equal(cols, allcols[lo:hi]), (lo, hi)
        assert np.array_equal(ci, info.reshape(nsims * nth, 2)[lo:hi])
    Hi, its = prob.implicit_H_batch(11, s0, s0 + nsims, theta)
    icol = Hi.transpose(0, 2, 1).reshape(nsims * nth, nth)
    for lo, hi in ((1, nth + 1), (nth - 1, 2 * nth + 1)):
        cols, ci = prob.implicit_H_columns(11, s0, lo, hi, theta)
        assert np.array_equal(cols, icol[lo:hi]) and np.array_equal(ci, its.reshape(-1)[lo:hi])
    with pytest.raises((M.MuseError, ValueError)):
        prob.fd_jacobian_columns(11, 0, 5, 2, theta, step)
    prob.close()


@pytest.mark.parametrize("split", [2, 4])
def test_element_split_hand_offs_under_uneven_load(gpu, M, O, split):
    """The tagged-granule exchange under load it was not tuned for: 1500 elements on clusters that each work through
    dozens of them (every buffer parity reused hundreds of times, epochs continuing across launches), uneven work
    (a warm restart in which two thirds of the elements converge without an iteration while the rest iterate), three
    launches per pattern bitwise equal, spot checks against the oracle."""
    N, nth, S = 9999, 3, 1500
    theta = np.array([0.4, 1.1, -0.3])
    prob = M.HipMuseProblem(None, model="funnel", ntheta=nth, N=N)
    prob.set_element_split(split)
    ref = None
    for _ in range(3):
        g, info = prob.map_and_score_batch(5, 0, S, theta, atol=1e-6)
        cur = (g.copy(), info.copy())
        ref = ref or cur
        assert np.array_equal(cur[0], ref[0]) and np.array_equal(cur[1], ref[1])
    assert np.all(info["status"] == 0)
    for k in (0, 749, 1499):
        go, zo, io = O.map_and_score_batch("funnel", N, 5, k, k + 1, theta, atol=1e-6, z0_mode=0)
        assert (info["iterations"][k], info["f_calls"][k]) == (io["iterations"][0], io["f_calls"][0])
        np.testing.assert_allclose(g[k], go[0], rtol=1e-10)
    # uneven: restart warm at the MAPs, but with a third of the slots overwritten by zeros (those iterate again)
    Z = prob.get_zhat(0, S)
    Z[::3] = 0.0
    prob.set_zhat(0, Z)
    ref2 = None
    for _ in range(3):
        prob.set_zhat(0, Z)
        g2, info2 = prob.map_and_score_batch(5, 0, S, theta, atol=1e-6, z0_mode=M.Z0_WARM)
        cur = (g2.copy(), info2.copy())
        ref2 = ref2 or cur
        assert np.array_equal(cur[0], ref2[0]) and np.array_equal(cur[1], ref2[1])
    assert np.all(info2["iterations"][1::3] == 0) and np.all(info2["iterations"][::3] >= 1)
    assert np.array_equal(g2[::3], g[::3])        # from zero again: the same path, the same bits
    np.testing.assert_allclose(g2, g, rtol=1e-9)
    prob.close()


@pytest.mark.gpu
def test_routing_of_five_to_eight_components_between_the_two_native_loops(gpu, M, monkeypatch, capfd):
    """muse_run_device's own routing (no MUSE_DEBUG_LOOP_ANY_NTHETA): with five to eight components the loop kernel runs where it was
    measured faster -- one problem per worker, and up to three in the all-register placement of 512 < N <= 4096 -- and the host loop
    elsewhere (N = 10^4 with two problems per worker); the same bits either way."""
    monkeypatch.delenv("MUSE_DEBUG_LOOP_ANY_NTHETA", raising=False)
    for N, nth, nsims, device in ((2048, 8, 512, True), (4096, 6, 700, True), (4096, 8, 1000, False), (10000, 8, 512, False), (10000, 8, 200, True), (10000, 4, 512, True), (10000, 4, 1900, False), (10000, 2, 1900, True)):
        x = np.sin(0.3 * np.arange(N)) + 0.5 * np.cos(1.7 * np.arange(N))
        prob = M.HipMuseProblem(x, model="funnel", ntheta=nth, prior=M.GaussianPrior(0.0, 3.0))
        kw = dict(nsims=nsims, maxsteps=3, theta_rtol=0.0, atol=1e-2, alpha=0.7)
        b = prob.run_muse(3, [0.2] * nth, device_loop=False, **kw)
        capfd.readouterr()
        prob.debug_flags(M.HipMuseProblem.DEBUG_RUN_TIMING)
        a = prob.run_muse(3, [0.2] * nth, device_loop=True, **kw)
        prob.debug_flags(0)
        assert ("[muse_run_device] launch call" in capfd.readouterr().err) == device, (N, nth, nsims)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3]) and a[4].tobytes() == b[4].tobytes()
        prob.close()
