"""Covariance estimators for J = cov(simulation scores): the `covariance_method` of get_J! (src/muse.jl:480, 495, 529).

The reference takes any `CovarianceEstimator` of CovarianceEstimation.jl (Project.toml: 0.2.7, not vendored under the reference's
tree) and defaults to `SimpleCovariance(corrected=true)`.  Restated here from the published formulas the package documents --
parity with the package itself is UNPINNED (nothing in this image runs Julia); what IS pinned: `SimpleCovariance` against numpy,
Ledoit-Wolf against scikit-learn's `ledoit_wolf`, OAS against scikit-learn's `oas` (tests/test_covariance.py).

    SimpleCovariance(corrected=True)                                  the sample covariance, 1/(n-1) or 1/n
    LinearShrinkage(target, shrinkage, corrected=False)               (1 - lam) S + lam F
        target     "DiagonalUnitVariance"     F = I
                   "DiagonalCommonVariance"   F = (tr S / p) I
                   "DiagonalUnequalVariance"  F = diag(S)
        shrinkage  a number in [0, 1], or
                   "lw"    Ledoit & Wolf (2004): lam = sum_ij Var^(s_ij) / sum_ij (s_ij - f_ij)^2, Var^(s_ij) = 1/n^2 sum_k (x_ki x_kj - s_ij)^2
                   "ss"    Schaefer & Strimmer (2005): the same on standardised data (the correlations are shrunk, the variances kept)
                   "rblw"  Chen, Wiesel, Eldar & Hero (2010), Rao-Blackwellised Ledoit-Wolf   (DiagonalCommonVariance only)
                   "oas"   the same paper's oracle-approximating shrinkage                      (DiagonalCommonVariance only)
    any callable G[nsims, ntheta] -> J[ntheta, ntheta]

Host-side and tiny (nsims x ntheta scores): not on the hot path.
"""
import numpy as np

_TARGETS = ("DiagonalUnitVariance", "DiagonalCommonVariance", "DiagonalUnequalVariance")


class SimpleCovariance:
    def __init__(self, corrected=False):
        self.corrected = bool(corrected)

    def __call__(self, G):
        G = np.atleast_2d(np.asarray(G, dtype=np.float64))
        n = G.shape[0]
        if n - (1 if self.corrected else 0) < 1:
            raise ValueError("SimpleCovariance: not enough samples")
        X = G - G.mean(axis=0)
        return X.T @ X / (n - 1 if self.corrected else n)

    def __repr__(self):
        return f"SimpleCovariance(corrected={self.corrected})"


class LinearShrinkage:
    def __init__(self, target="DiagonalUnitVariance", shrinkage="lw", corrected=False):
        target = str(target).replace("()", "")
        if target not in _TARGETS:
            raise ValueError(f"LinearShrinkage: target {target!r} is not one of {_TARGETS}")
        if isinstance(shrinkage, str):
            shrinkage = shrinkage.lstrip(":")
            if shrinkage not in ("lw", "ss", "rblw", "oas"):
                raise ValueError(f"LinearShrinkage: shrinkage {shrinkage!r} is not a number or one of lw, ss, rblw, oas")
            if shrinkage in ("rblw", "oas") and target != "DiagonalCommonVariance":
                raise ValueError(f"LinearShrinkage: {shrinkage} is defined for the DiagonalCommonVariance target")
        elif not 0.0 <= float(shrinkage) <= 1.0:
            raise ValueError("LinearShrinkage: a fixed shrinkage lies in [0, 1]")
        self.target, self.shrinkage, self.corrected = target, shrinkage, bool(corrected)
        self.lam = None          # the intensity of the last call

    def _target(self, S):
        p = S.shape[0]
        if self.target == "DiagonalUnitVariance":
            return np.eye(p)
        if self.target == "DiagonalCommonVariance":
            return np.trace(S) / p * np.eye(p)
        return np.diag(np.diag(S))

    def _lw(self, X, S):
        n = X.shape[0]
        F = self._target(S)
        # Var^(s_ij) = 1/n^2 sum_k (x_ki x_kj - s_ij)^2 with s the 1/n covariance: sum_k (x_ki x_kj)^2 / n^2 - s_ij^2 / n
        S0 = X.T @ X / n
        var = ((X * X).T @ (X * X)) / n ** 2 - S0 * S0 / n
        if self.target == "DiagonalUnequalVariance":        # the target follows the diagonal: only the off-diagonal entries are shrunk
            var = var - np.diag(np.diag(var))
        den = np.sum((S - F) ** 2)
        return 0.0 if den <= 0.0 else float(np.clip(np.sum(var) / den, 0.0, 1.0))

    def __call__(self, G):
        G = np.atleast_2d(np.asarray(G, dtype=np.float64))
        n, p = G.shape
        if n < 2:
            raise ValueError("LinearShrinkage: not enough samples")
        X = G - G.mean(axis=0)
        S = X.T @ X / (n - 1 if self.corrected else n)
        sh = self.shrinkage
        if not isinstance(sh, str):
            lam = float(sh)
        elif sh == "lw":
            lam = self._lw(X, S)
        elif sh == "ss":
            sd = np.sqrt(np.diag(S))
            sd = np.where(sd > 0.0, sd, 1.0)
            Xs = X / sd
            Rs = Xs.T @ Xs / (n - 1 if self.corrected else n)
            lam = self._lw(Xs, Rs)
            self.lam = lam
            R = (1.0 - lam) * Rs + lam * self._target(Rs)
            return R * np.outer(sd, sd)
        else:
            trS, trS2 = np.trace(S), np.sum(S * S)
            if sh == "rblw":
                num = (n - 2.0) / n * trS2 + trS ** 2
                den = (n + 2.0) * (trS2 - trS ** 2 / p)
            else:  # oas
                num = (1.0 - 2.0 / p) * trS2 + trS ** 2
                den = (n + 1.0 - 2.0 / p) * (trS2 - trS ** 2 / p)
            lam = 1.0 if den <= 0.0 else float(min(num / den, 1.0))
        self.lam = lam
        return (1.0 - lam) * S + lam * self._target(S)

    def __repr__(self):
        return f"LinearShrinkage({self.target}, {self.shrinkage!r}, corrected={self.corrected})"


def as_covariance_method(m):
    """The estimator get_J_ applies to the scores: an instance of the classes above, a callable G -> J, or a name
    ("simple_corrected" -- the reference's default, src/muse.jl:495 --, "simple", "lw", "ss", "oas", "rblw")."""
    if m is None or m == "simple_corrected":
        return SimpleCovariance(corrected=True)
    if callable(m):
        return m
    if isinstance(m, str):
        name = m.lstrip(":")
        if name == "simple":
            return SimpleCovariance(corrected=False)
        if name in ("lw", "ss"):
            return LinearShrinkage("DiagonalUnequalVariance" if name == "ss" else "DiagonalCommonVariance", name)
        if name in ("oas", "rblw"):
            return LinearShrinkage("DiagonalCommonVariance", name)
    raise ValueError(f"covariance_method {m!r}: an estimator of museinference_jl_amd.covariance, a callable scores -> J, or one of "
                     "'simple_corrected', 'simple', 'lw', 'ss', 'oas', 'rblw'")
