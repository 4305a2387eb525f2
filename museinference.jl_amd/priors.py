"""log-prior objects for logPriorθ (reference src/interface.jl:107-121, src/simple.jl:69-71,93).

muse! needs the prior's gradient and Hessian in theta (src/muse.jl:184,207,539).  The reference gets
them from ForwardDiff; here a prior object supplies them analytically (GaussianPrior, FlatPrior) or,
for an arbitrary callable, by central differences (CallablePrior).
"""
import numpy as np


class FlatPrior:
    """logPriorθ = 0, the interface default (src/interface.jl:120-121)."""

    def logpdf(self, theta):
        return 0.0

    def grad(self, theta):
        return np.zeros_like(np.asarray(theta, dtype=np.float64))

    def hess(self, theta):
        n = np.asarray(theta).size
        return np.zeros((n, n))


class GaussianPrior:
    """Independent normal prior, e.g. the funnel's  -θ²/(2·3²)  (src/simple.jl:69-71)."""

    def __init__(self, mean=0.0, sigma=3.0):
        self.mean = mean
        self.sigma = sigma

    def _ms(self, theta):
        theta = np.asarray(theta, dtype=np.float64)
        return theta, np.broadcast_to(np.asarray(self.mean, dtype=np.float64), theta.shape), \
            np.broadcast_to(np.asarray(self.sigma, dtype=np.float64), theta.shape)

    def logpdf(self, theta):
        t, m, s = self._ms(theta)
        return float(np.sum(-((t - m) ** 2) / (2 * s**2)))

    def grad(self, theta):
        t, m, s = self._ms(theta)
        return -(t - m) / s**2

    def hess(self, theta):
        t, m, s = self._ms(theta)
        return np.diag(-1.0 / s**2 * np.ones_like(t))


class CallablePrior:
    """Wraps a plain function theta -> log prior; derivatives by central differences."""

    def __init__(self, fn, step=1e-4):
        self.fn = fn
        self.step = step

    def logpdf(self, theta):
        return float(self.fn(np.asarray(theta, dtype=np.float64)))

    def grad(self, theta):
        t = np.asarray(theta, dtype=np.float64)
        g = np.zeros_like(t)
        for i in range(t.size):
            e = np.zeros_like(t)
            e[i] = self.step
            g[i] = (self.logpdf(t + e) - self.logpdf(t - e)) / (2 * self.step)
        return g

    def hess(self, theta):
        t = np.asarray(theta, dtype=np.float64)
        n = t.size
        H = np.zeros((n, n))
        for i in range(n):
            e = np.zeros_like(t)
            e[i] = self.step
            H[:, i] = (self.grad(t + e) - self.grad(t - e)) / (2 * self.step)
        return 0.5 * (H + H.T)


def as_prior(p):
    if p is None:
        return FlatPrior()
    if all(hasattr(p, a) for a in ("logpdf", "grad", "hess")):
        return p
    if callable(p):
        return CallablePrior(p)
    raise TypeError("prior must be None, a callable or an object with logpdf/grad/hess")
