"""SimpleMuseProblem (src/simple.jl:4-12, 79-95): a problem given as CLOSURES, differentiated by AD.

    prob = TorchMuseProblem(x, sample_x_z, logLike, logPrior=None, device="cuda")
        sample_x_z(generator, theta) -> (x, z)      torch tensors on `device`, drawn with the torch.Generator it is given
        logLike(x, z, theta)         -> 0-d tensor  log P(x, z | theta), differentiable in z and theta (theta: a float64 tensor)
        logPrior(theta)              -> 0-d tensor  (default: flat)

grad_z and grad_theta come from torch.autograd -- the reference uses ForwardDiff / Zygote (src/simple.jl:84-85) -- and the MAP from
the interface's default `ẑ_at_θ` (src/interface.jl:140-166; optim.py: L-BFGS + HagerZhang on the device's tensors).  This is the
GENERAL front-end: any logLike, any coupling between the elements, any number of parameters -- at the price of one autograd pass and
a dozen small torch kernels per evaluation, simulation after simulation.  A model of the elementwise families belongs in a header
(models.ElementwiseModel: hand-written, or generated from its terms by symbolic.py) behind HipMuseProblem, where a whole batch of
simulations is ONE launch of the HIP solver; this class is what runs the models those families do not hold.  Nothing routes from
one to the other: which one runs is the caller's choice of class.

Everything of the host driver applies (muse, get_J_, get_H_ by finite differences, transforms, checkpoints): the driver sees an
AbstractMuseProblem.
"""
import numpy as np

from .problem import AbstractMuseProblem, SimRng, UnTransformedθ
from . import optim


class TorchMuseProblem(AbstractMuseProblem):
    def __init__(self, x, sample_x_z, logLike, logPrior=None, device=None, dtype=None):
        import torch
        self._torch = torch
        self.device = torch.device(device) if device is not None else (x.device if torch.is_tensor(x) else torch.device("cpu"))
        self.dtype = dtype or torch.float64
        self.x = None if x is None else self._t(x)
        self._sample, self._logLike, self._logPrior = sample_x_z, logLike, logPrior

    # -- tensors in, numpy for theta-sized results out (the driver's algebra is numpy)
    def _t(self, v):
        torch = self._torch
        return v.to(self.device, self.dtype) if torch.is_tensor(v) else torch.as_tensor(np.asarray(v, dtype=np.float64), device=self.device).to(self.dtype)

    def _theta(self, theta, grad=False):
        t = self._torch.as_tensor(np.atleast_1d(np.asarray(theta, dtype=np.float64)), device=self.device).to(self.dtype)
        return t.requires_grad_(True) if grad else t

    def sample_x_z(self, rng, theta):
        """A stream per (master seed, simulation index), never advanced by the caller (src/util.jl:87-92): the generator handed to the
        closure is seeded from the pair, so a simulation's draw depends on nothing else."""
        torch = self._torch
        gen = torch.Generator(device=self.device)
        gen.manual_seed(self._stream(rng if isinstance(rng, SimRng) else SimRng(int(rng), 0)))
        with torch.no_grad():
            x, z = self._sample(gen, self._theta(theta))
        return self._t(x), self._t(z)

    def logLike_and_grad_z_logLike(self, x, z, theta):
        torch = self._torch
        zt = self._t(z).detach().clone().requires_grad_(True)
        with torch.enable_grad():
            f = self._logLike(self._t(x), zt, self._theta(theta))
            g, = torch.autograd.grad(f, zt)
        return float(f.detach()), g

    def grad_theta_logLike(self, x, z, theta, theta_space=UnTransformedθ):
        torch = self._torch
        th = self.inv_transform_theta(theta) if theta_space is not UnTransformedθ else theta
        tt = self._theta(th, grad=True)
        with torch.enable_grad():
            f = self._logLike(self._t(x), self._t(z).detach(), tt)
            g, = torch.autograd.grad(f, tt, allow_unused=True)
        g = np.zeros(tt.numel()) if g is None else g.detach().cpu().numpy().astype(np.float64)
        if theta_space is not UnTransformedθ:       # chain rule through the transform, by central differences (as check_self_consistency)
            g = self._jac_inv_transform(np.atleast_1d(np.asarray(theta, dtype=np.float64))).T @ g
        return g

    def _jac_inv_transform(self, theta_t, step=1e-6):
        n = theta_t.size
        J = np.zeros((n, n))
        for j in range(n):
            e = np.zeros(n)
            e[j] = step
            J[:, j] = (np.atleast_1d(self.inv_transform_theta(theta_t + e)) - np.atleast_1d(self.inv_transform_theta(theta_t - e))) / (2 * step)
        return J

    def zhat_at_theta(self, x, z0, theta, grad_z_logLike_atol=1e-2):
        """ẑ_at_θ, the interface's default (src/interface.jl:140-166): minimise -logLike over z from z0, g_tol = the tolerance."""
        torch = self._torch
        xt, tt = self._t(x), self._theta(theta)

        def fg(z):
            zt = z.detach().clone().requires_grad_(True)
            with torch.enable_grad():
                f = -self._logLike(xt, zt, tt)
                g, = torch.autograd.grad(f, zt)
            return float(f.detach()), g
        z, info = optim.lbfgs(fg, self._t(z0).reshape(-1), grad_z_logLike_atol)
        rec = np.zeros((), dtype=_info_dtype())
        for k in ("iterations", "f_calls", "status"):
            rec[k] = info[k]
        rec["f_min"], rec["gnorm"] = info["f_min"], info["gnorm"]
        return z.reshape(self._t(z0).shape), rec

    def zhat_guess_from_truth(self, x, z, theta):
        return self._torch.zeros_like(self._t(z))

    # -- get_H! by implicit differentiation (src/muse.jl:335-405), every derivative by autograd as the reference takes them by nested AD
    def implicit_H_batch(self, seed, sim_begin, sim_end, theta0, atol=1e-1, cg_maxiter=100, cg_reltol=1.4901161193847656e-08):
        """Per simulation H = H1 - dFdθᵀ A⁻¹ dFdθ1: H1 the Jacobian of the score through the DRAW x(θ) at fixed ẑ, dFdθ = ∂θ ∇z logLike,
        dFdθ1 = ∂θ ∇z logLike(x(θ), ẑ, θ₀), A the Hessian of logLike in z applied by double backward, A⁻¹ by conjugate gradients (reltol
        sqrt(eps), maxiter 100: IterativeSolvers' defaults).  The draw is differentiated through the closure itself (the generator
        re-seeded: the same random numbers at every θ), so sample_x_z must be written with differentiable torch operations.
        Returns (Hs [n, nθ, nθ], CG iteration counts [n, nθ])."""
        torch = self._torch
        th0 = self._theta(theta0)
        nt = th0.numel()
        Hs, its = [], []
        jac = torch.autograd.functional.jacobian
        for sim in range(int(sim_begin), int(sim_end)):
            rng = SimRng(int(seed), sim)
            x, z = self.sample_x_z(rng, theta0)
            zh, _ = self.zhat_at_theta(x, self.zhat_guess_from_truth(x, z, theta0), theta0, atol)
            zh = zh.detach()

            def x_of(th):          # the draw at th, the simulation's own random numbers
                gen = torch.Generator(device=self.device)
                gen.manual_seed(self._stream(rng))
                return self._sample(gen, th)[0].to(self.dtype)

            def grad_z(xx, zz, th):
                zz = zz if zz.requires_grad else zz.detach().clone().requires_grad_(True)
                return torch.autograd.grad(self._logLike(xx, zz, th), zz, create_graph=True)[0]

            def score(xx, th_eval):
                tt = th_eval.detach().clone().requires_grad_(True)
                return torch.autograd.grad(self._logLike(xx, zh, tt), tt, create_graph=True)[0]
            with torch.enable_grad():
                H1 = jac(lambda th: score(x_of(th), th0), th0).reshape(nt, nt)
                dF = jac(lambda th: grad_z(x, zh, th), th0).reshape(-1, nt)
                dF1 = jac(lambda th: grad_z(x_of(th), zh, th0), th0).reshape(-1, nt)
                zz = zh.clone().requires_grad_(True)
                g = torch.autograd.grad(self._logLike(x, zz, th0), zz, create_graph=True)[0].reshape(-1)

                def A(w):          # Hessian of logLike in z at the MAP, applied
                    return torch.autograd.grad(g, zz, w.reshape(zz.shape), retain_graph=True)[0].reshape(-1)
                cols, n_it = [], []
                for j in range(nt):
                    y, k = _cg(lambda w: -A(w), -dF1[:, j].detach(), cg_maxiter, cg_reltol)     # (-A is positive definite at a maximum)
                    cols.append(y)
                    n_it.append(k)
            H = H1.detach() - dF.detach().T @ torch.stack(cols, dim=1)
            Hs.append(H.cpu().numpy().astype(np.float64))
            its.append(n_it)
        return np.array(Hs).reshape(-1, nt, nt), np.array(its, dtype=np.int32).reshape(-1, nt)

    @staticmethod
    def _stream(rng):
        return (int(rng.seed) * 0x9E3779B97F4A7C15 + int(rng.sim) * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) % (1 << 63)

    # -- prior: the closure, differentiated (the reference: ForwardDiff, src/muse.jl:184,207,539)
    def logPrior_theta(self, theta, theta_space=UnTransformedθ):
        if self._logPrior is None:
            return 0.0
        th = self.inv_transform_theta(theta) if theta_space is not UnTransformedθ else theta
        return float(self._logPrior(self._theta(th)))

    def grad_logPrior_theta(self, theta, theta_space=UnTransformedθ):
        n = np.atleast_1d(np.asarray(theta)).size
        if self._logPrior is None:
            return np.zeros(n)
        if theta_space is not UnTransformedθ:
            return _fd_grad(lambda t: self.logPrior_theta(t, theta_space), np.atleast_1d(np.asarray(theta, dtype=np.float64)))
        torch = self._torch
        tt = self._theta(theta, grad=True)
        with torch.enable_grad():
            g, = torch.autograd.grad(self._logPrior(tt), tt, allow_unused=True)
        return np.zeros(n) if g is None else g.detach().cpu().numpy().astype(np.float64)

    def hess_logPrior_theta(self, theta, theta_space=UnTransformedθ):
        n = np.atleast_1d(np.asarray(theta)).size
        if self._logPrior is None:
            return np.zeros((n, n))
        t0 = np.atleast_1d(np.asarray(theta, dtype=np.float64))
        if theta_space is not UnTransformedθ:
            H = np.zeros((n, n))
            for j in range(n):
                e = np.zeros(n)
                e[j] = 1e-5
                H[:, j] = (self.grad_logPrior_theta(t0 + e, theta_space) - self.grad_logPrior_theta(t0 - e, theta_space)) / 2e-5
            return 0.5 * (H + H.T)
        torch = self._torch
        H = torch.autograd.functional.hessian(lambda t: self._logPrior(t), self._theta(t0))
        return H.detach().cpu().numpy().astype(np.float64).reshape(n, n)


def _cg(A, b, maxiter, reltol):
    """Conjugate gradients for A y = b from y = 0 (A symmetric positive definite, given as a function): (y, iterations); stops at
    |r| <= reltol |b| or after maxiter iterations."""
    y = b.new_zeros(b.shape)
    r = b.clone()
    p = r.clone()
    rs = float(r.dot(r))
    stop = (reltol ** 2) * rs
    k = 0
    while k < maxiter and rs > stop and rs > 0.0:
        Ap = A(p)
        alpha = rs / float(p.dot(Ap))
        y = y + alpha * p
        r = r - alpha * Ap
        rs_new = float(r.dot(r))
        p = r + (rs_new / rs) * p
        rs = rs_new
        k += 1
    return y, k


def _fd_grad(f, t, step=1e-6):
    g = np.zeros(t.size)
    for j in range(t.size):
        e = np.zeros(t.size)
        e[j] = step
        g[j] = (f(t + e) - f(t - e)) / (2 * step)
    return g


def _info_dtype():
    from . import _capi
    return _capi.INFO_DTYPE
