"""Finite-difference methods for get_H! -- the part of FiniteDifferences.jl (compat 0.12.20, reference Project.toml:40; not
vendored in the reference) that `fdm = central_fdm(3,1)` / `fdm(f, x[, step])` at src/muse.jl:300 and src/util.jl:13 reach,
restated from the package's published algorithm (Fornberg-type coefficients from the Vandermonde system, and the step
that minimises the bound  C1 h^-Q + C2 h^(P-Q)  on round-off + truncation error, with |f^(P)| estimated by a second,
unadapted method of order P + 2).  As recalled -- FiniteDifferences cannot be run here -- and marked so in DESIGN.md §5 / HISTORY.md §4.

    m = central_fdm(p, q)                   p grid points, q-th derivative, adapt = 1, condition = 10, factor = 1
    m.grid, m.coefs                         e.g. central_fdm(3, 1): (-1, 0, 1), (-1/2, 0, 1/2)
    step = m.estimate_step(fvals, x)        fvals(offsets) -> values of f at x + offsets  ([G] or [G, k] array)
    m.estimate(values, step)                sum(coefs * values) / step**q

The engine evaluates f in batches (HipMuseProblem.fd_values_columns); this module only holds the scalar algebra.
"""
import functools
import math
from fractions import Fraction

import numpy as np


def _coefs(grid, q):
    """Coefficients c with sum_g c_g g^i = q! delta_{iq}, i = 0..p-1, solved exactly (rationals) and rounded once."""
    c = _coefs_exact(tuple(int(g) for g in grid), int(q))
    c.flags.writeable = False      # one array per (grid, q) for the whole process: the rational solve costs ~0.3 ms, and a
    return c                       # muse(get_covariance=True) at configs[1] is ~0.3 ms altogether without it


@functools.lru_cache(maxsize=None)
def _coefs_exact(grid, q):
    p = len(grid)
    A = [[Fraction(g) ** i for g in grid] + [Fraction(math.factorial(q)) if i == q else Fraction(0)] for i in range(p)]
    for col in range(p):
        piv = next(r for r in range(col, p) if A[r][col] != 0)
        A[col], A[piv] = A[piv], A[col]
        d = A[col][col]
        A[col] = [v / d for v in A[col]]
        for r in range(p):
            if r != col and A[r][col] != 0:
                f = A[r][col]
                A[r] = [a - f * b for a, b in zip(A[r], A[col])]
    return np.array([float(A[i][p]) for i in range(p)])


def _central_grid(p):
    """_default_mid: -(p-1)/2 .. (p-1)/2 for odd p; for even p the integers -p/2 .. p/2 without 0."""
    if p % 2 == 1:
        return list(range(-(p // 2), p // 2 + 1))
    return list(range(-(p // 2), 0)) + list(range(1, p // 2 + 1))


class FiniteDifferenceMethod:
    def __init__(self, grid, q, adapt=1, condition=10.0, factor=1.0, max_range=math.inf):
        self.grid = [int(g) for g in grid]
        self.p, self.q = len(self.grid), int(q)
        if not 0 <= self.q < self.p:
            raise ValueError("order of the method must be strictly greater than the order of the derivative")
        self.condition, self.factor, self.max_range = float(condition), float(factor), float(max_range)
        self.coefs = _coefs(self.grid, self.q)
        g = np.array(self.grid, dtype=np.float64)
        # estimates of the derivative one grid step to the left of, at, and to the right of x from the SAME function values
        if all(v >= 0 for v in self.grid):
            self.coefs_neighbourhood = [self.coefs, _coefs([v - 1 for v in self.grid], self.q)]
        elif all(v <= 0 for v in self.grid):
            self.coefs_neighbourhood = [self.coefs, _coefs([v + 1 for v in self.grid], self.q)]
        else:
            self.coefs_neighbourhood = [_coefs([v - 1 for v in self.grid], self.q), self.coefs,
                                        _coefs([v + 1 for v in self.grid], self.q)]
        self.grad_magnitude_mult = float(np.sum(np.abs(self.coefs * g ** self.p))) / math.factorial(self.p)
        self.f_error_mult = float(np.sum(np.abs(self.coefs)))
        # adapt >= 1: |f^(p)| is estimated with a method of order p + 2 for the p-th derivative, itself adapted adapt - 1 times
        self.bound_estimator = central_fdm(self.p + 2, self.p, adapt=adapt - 1, condition=condition, factor=factor,
                                           max_range=max_range) if adapt >= 1 else None

    def __repr__(self):
        return f"FiniteDifferenceMethod(grid={self.grid}, q={self.q}, coefs={self.coefs.tolist()})"

    # -- step selection
    def _step_acc(self, grad_magnitude, f_error):
        P, Q = self.p, self.q
        c1 = f_error * self.f_error_mult * self.factor
        c2 = grad_magnitude * self.grad_magnitude_mult
        step = (Q / (P - Q) * (c1 / c2)) ** (1.0 / P)
        return step, c1 * step ** (-Q) + c2 * step ** (P - Q)

    def default_step(self):
        """The heuristic step: |f^(P)| taken as `condition`, f's error as eps(Float64)."""
        return self._step_acc(self.condition, np.finfo(np.float64).eps)[0]

    def _limit(self, step):
        step_max = self.max_range / max(abs(v) for v in self.grid)
        if step > step_max:
            step = step_max
        return min(step, 1000.0 * self.default_step())

    def magnitudes(self, fvals, x=0.0):
        """(max |d^q f| over the neighbourhood estimates, max |f|) of this method's own evaluation (_estimate_magnitudes)."""
        step = self.estimate_step(fvals, x)
        F = np.asarray(fvals(step * np.array(self.grid, dtype=np.float64)), dtype=np.float64)
        return self.magnitudes_from_values(F, step)

    def magnitudes_from_values(self, F, step):
        F = np.asarray(F, dtype=np.float64).reshape(self.p, -1)
        grads = [np.tensordot(c, F, axes=(0, 0)) / step ** self.q for c in self.coefs_neighbourhood]
        return max(float(np.max(np.abs(g))) for g in grads), float(np.max(np.abs(F)))

    def step_from_magnitudes(self, grad_magnitude, f_magnitude):
        if grad_magnitude == 0.0 or f_magnitude == 0.0 or not (np.isfinite(grad_magnitude) and np.isfinite(f_magnitude)):
            step = self.default_step()
        else:
            step = self._step_acc(grad_magnitude, float(np.spacing(f_magnitude)))[0]
        return self._limit(step)

    def estimate_step(self, fvals, x=0.0):
        if self.bound_estimator is None:
            return self._limit(self.default_step())
        gm, fm = self.bound_estimator.magnitudes(fvals, x)
        return self.step_from_magnitudes(gm, fm)

    # -- the estimate
    def estimate(self, F, step):
        """sum_g coefs[g] F[g] / step^q   (F [p] or [p, k])"""
        return np.tensordot(self.coefs, np.asarray(F, dtype=np.float64), axes=(0, 0)) / step ** self.q

    def __call__(self, f, x=0.0, step=None):
        """fdm(f, x[, step]): f maps a float to a float or an array."""
        fvals = lambda offs: np.array([np.asarray(f(x + o), dtype=np.float64) for o in offs])
        if step is None:
            step = self.estimate_step(fvals, x)
        return self.estimate(fvals(step * np.array(self.grid, dtype=np.float64)), step)


def central_fdm(p, q, adapt=1, condition=10.0, factor=1.0, max_range=math.inf):
    """central_fdm(p, q; adapt = 1, condition = 10, factor = 1, max_range = Inf)"""
    return FiniteDifferenceMethod(_central_grid(int(p)), q, adapt=adapt, condition=condition, factor=factor, max_range=max_range)


@functools.lru_cache(maxsize=64)
def _from_spelling(spec):
    """(a method holds no state that its calls change: one instance per spelling serves every get_H!)"""
    import re
    m = re.fullmatch(r"\s*central_fdm\(\s*(\d+)\s*,\s*(\d+)\s*\)\s*", spec)
    return central_fdm(int(m.group(1)), int(m.group(2))) if m else None


def as_fdm(spec):
    """A FiniteDifferenceMethod from an instance or from the spelling "central_fdm(p,q)"."""
    if isinstance(spec, FiniteDifferenceMethod):
        return spec
    if isinstance(spec, str):
        m = _from_spelling(spec)
        if m is not None:
            return m
    raise ValueError(f"fdm must be a FiniteDifferenceMethod or 'central_fdm(p,q)', got {spec!r}")
