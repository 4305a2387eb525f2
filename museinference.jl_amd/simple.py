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
        seed, sim = (rng.seed, rng.sim) if isinstance(rng, SimRng) else (int(rng), 0)
        gen = torch.Generator(device=self.device)
        gen.manual_seed((int(seed) * 0x9E3779B97F4A7C15 + int(sim) * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) % (1 << 63))
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
