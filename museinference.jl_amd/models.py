"""User-supplied elementwise models: SimpleMuseProblem's closures (src/simple.jl:79-95) as compiled code.

The reference builds a problem from `sample_x_z`, `logLike` and `logPrior` closures and differentiates them by AD.  Here a
model is a C header with three functions (include/muse_model.h states the contract and the family:
-logLike = 1/2 sum_i [A(x_i, z_i) + e^-theta_k B(x_i, z_i)] + 1/2 sum_k n_k theta_k) that hipcc compiles into an engine
library of its own -- every placement of the solver kernel, the sampler, the finite-difference and multi-map launches, the
native muse! loop, behind the same C ABI:

    model = ElementwiseModel.from_source("cubic", '''
        #include "muse_model.h"
        #define MUSE_MODEL_NAME "cubic"
        MUSE_MODEL_FN void   muse_model_sample(double sd, double n1, double n2, double* z, double* x, long i) { ... }
        MUSE_MODEL_FN double muse_model_grad(double iv, double x, double z, double* acc, long i) { ... }
        MUSE_MODEL_FN double muse_model_score_term(double x, double z, long i) { ... }
    ''', constants={"P": spectrum})        # optional: per-element tables the functions read as P(i) -- what a closure captures
    prob = HipMuseProblem(x, model=model, ntheta=2, prior=GaussianPrior(0, 3))
    result = muse(prob, [0.0, 0.0], get_covariance=True)          # get_H! by finite differences

`ElementwiseModel.packaged("cubic")` is the example shipped in museinference.jl_amd/models/cubic.h.
"""
import hashlib
import os
import re

from . import build as _build


class ElementwiseModel:
    def __init__(self, name, header):
        if not re.fullmatch(r"[A-Za-z][A-Za-z0-9_]*", name or ""):
            raise ValueError(f"model name {name!r}: letters, digits and underscores, starting with a letter")
        self.name = name
        self.header = os.path.abspath(header)
        if not os.path.exists(self.header):
            raise FileNotFoundError(self.header)
        self._library = None

    @classmethod
    def packaged(cls, name):
        """One of the example models in museinference.jl_amd/models/."""
        return cls(name, os.path.join(_build.MODELS_DIR, name + ".h"))

    @classmethod
    def from_source(cls, name, source, directory=None, constants=None, runtime_constants=None):
        """Write `source` to <directory>/<name>_<hash>.h (default: museinference.jl_amd/models/user/, or $MUSE_MODEL_DIR/headers) and wrap it; the
        library is named after name and hash, so that an edited source gets a library of its own.

        constants: {"P": array of N doubles, ...} -- per-element constants the model's functions look up by the element index
        they are given (what a closure of the reference's SimpleMuseProblem would capture: a known spectrum, a noise-variance
        map, a mask).  Each becomes a table compiled into the header and an accessor `double P(long i)`; the model is then
        built for that N (MUSE_MODEL_N: the engine refuses a context of another size)."""
        if not re.fullmatch(r"[A-Za-z][A-Za-z0-9_]*", str(name)):   # before any path is formed from it
            raise ValueError(f"model name {name!r}: letters, digits and underscores, starting with a letter")
        directory = directory or (os.path.join(_build.model_out_dir(), "headers") if os.environ.get("MUSE_MODEL_DIR")
                                  else os.path.join(_build.MODELS_DIR, "user"))
        os.makedirs(directory, exist_ok=True)
        if constants and runtime_constants:
            raise ValueError("give the per-element constants either as compiled tables (constants=) or at run time (runtime_constants=)")
        if constants:
            source = cls._tables(constants) + source
        if runtime_constants:
            # run-time constants (include/muse_model.h, muse_const): the values are set on the problem -- HipMuseProblem(...,
            # constants={"P": array}) / set_constants -- and can be replaced without touching the library
            names = list(runtime_constants)
            for k in names:
                if not re.fullmatch(r"[A-Za-z][A-Za-z0-9_]*", str(k)):
                    raise ValueError(f"runtime_constants: {k!r} is not a C identifier")
            if not 1 <= len(names) <= 4:
                raise ValueError("runtime_constants: between 1 and 4 names (MUSE_MODEL_MAX_CONST)")
            head = [f"#define MUSE_MODEL_NCONST {len(names)}", '#include "muse_model.h"']
            head += [f"MUSE_MODEL_FN double {k}(long i) {{ return muse_const({j}, i); }}" for j, k in enumerate(names)]
            source = "\n".join(head) + "\n" + source
        tag = hashlib.sha256(source.encode()).hexdigest()[:10]
        path = os.path.join(directory, f"{name}_{tag}.h")
        if not os.path.exists(path):
            with open(path, "w") as f:
                f.write(source)
        m = cls(name, path)
        m._libname = f"{name}_{tag}"
        m.runtime_constants = list(runtime_constants) if runtime_constants else []
        return m

    @classmethod
    def from_expressions(cls, name, A, B, z, x, directory=None, constants=None, runtime_constants=None):
        """A model of the one-parameter family from its TERMS -- A(x, z), B(x, z) of -logLike = 1/2 sum [A + e^-theta B] + 1/2 sum n theta,
        and the draw z(sd, n1, n2), x(z, sd, n1, n2) -- as expressions (strings in x, z, sd, n1, n2 and the names of the per-element
        constants, or sympy expressions): every derivative the engine needs, the second derivatives of the implicit-differentiation
        get_H! included, is formed symbolically (museinference_jl_amd.symbolic) -- the counterpart of the AD the reference applies to
        a SimpleMuseProblem's closures (src/simple.jl:84-85).  `ElementwiseModel.from_expressions("cubic_gen",
        A="(x - (z + z**3/10))**2", B="z**2", z="sd*n1", x="z + z**3/10 + n2")` is models/cubic.h."""
        from .symbolic import header_from_expressions
        names = list(constants or {}) + list(runtime_constants or [])
        return cls.from_source(name, header_from_expressions(name, A, B, z, x, constant_names=names), directory=directory,
                               constants=constants, runtime_constants=runtime_constants)

    @classmethod
    def from_pair_expressions(cls, name, coefs, C, o, z, x, directory=None):
        """A model of the TWO-parameter family (include/muse_model.h, MUSE_MODEL_PAIR: block k's parameters a = theta[k],
        b = theta[K + k]) from its terms: the coefficients c0 .. c3 as expressions in a, b (exp allowed there), the block's constant
        C(a, b), the element's objective term o(c; x, z) and the draw z(c0, c1, n1, n2), x(z, c0, c1, n1, n2) -- gradient and score by
        symbolic differentiation (museinference_jl_amd.symbolic.pair_header_from_expressions).  models/normal_mean_var.h is
        `from_pair_expressions("nmv", coefs=["a", "exp(b/2)", "exp(-b)"], C="b", o="(x - z)**2 + c2*(z - c0)**2", z="c0 + c1*n1", x="z + n2")`."""
        from .symbolic import pair_header_from_expressions
        return cls.from_source(name, pair_header_from_expressions(name, coefs, C, o, z, x), directory=directory)

    @staticmethod
    def _tables(constants):
        import numpy as np
        arrays = {k: np.ascontiguousarray(np.asarray(v, dtype=np.float64)).reshape(-1) for k, v in constants.items()}
        sizes = {a.size for a in arrays.values()}
        if len(sizes) != 1:
            raise ValueError("constants: every array needs one entry per element (the same N)")
        n = sizes.pop()
        out = ['#include "muse_model.h"', f"#define MUSE_MODEL_N {n}"]
        for k, a in arrays.items():
            if not re.fullmatch(r"[A-Za-z][A-Za-z0-9_]*", k):
                raise ValueError(f"constants: {k!r} is not a C identifier")
            if not np.all(np.isfinite(a)):
                raise ValueError(f"constants: {k} has non-finite entries")
            body = ",\n".join(", ".join(repr(float(v)) for v in a[j:j + 8]) for j in range(0, n, 8))
            # entry N serves the zero pad element of an odd-length vector and the phantom slots beyond it (x = z = 0 there)
            out.append(f"static const double {k}_table[{n + 1}] = {{\n{body},\n1.0}};")
            out.append(f"MUSE_MODEL_FN double {k}(long i) {{ return {k}_table[i < MUSE_MODEL_N ? i : MUSE_MODEL_N]; }}")
        return "\n".join(out) + "\n"

    @property
    def pair(self):
        """True for a header of the two-parameter family (include/muse_model.h: `#define MUSE_MODEL_PAIR`): ntheta = 2 K, block k's
        parameters are theta[k] and theta[K + k]."""
        if not hasattr(self, "_pair"):
            text = re.sub(r"/\*.*?\*/", "", open(self.header).read(), flags=re.S)
            self._pair = re.search(r"^\s*#\s*define\s+MUSE_MODEL_PAIR\b", text, flags=re.M) is not None
        return self._pair

    @property
    def library_name(self):
        return getattr(self, "_libname", self.name)

    def library(self, force=False):
        """Path of the model's engine library, compiled on first use (hipcc, ~1 min; cross-compiles without a GPU)."""
        if self._library is None or force:
            self._library = _build.build_model_library(self.header, self.library_name, force=force)
        return self._library

    def __repr__(self):
        return f"ElementwiseModel({self.name!r}, {self.header!r})"


def check_model_consistency(prob, theta, rng=0, n_probe=6, step=1e-5, rtol=2e-5):
    """What AD guarantees in the reference (src/simple.jl:84-85: the gradients ARE derivatives of the user's logLike) has to be
    checked for a hand-written header: through the problem's own per-simulation operators, at a draw (x, z) ~ P(x, z | theta),

        grad_z logLike           against central differences of logLike in n_probe elements of z,
        grad_theta logLike       against central differences of logLike in every theta_k
                                 (the family identity of include/muse_model.h: the score assembled from B is the derivative),

        the second derivatives of a header with MUSE_MODEL_SECOND (what the implicit-differentiation get_H! builds on) against
                                 central differences of the header's own first-order functions (_check_second, "second"),

    and returns {"grad_z": worst residual, "grad_theta": worst residual, "noise_floor": ...} -- residuals relative to the larger
    of the gradient's size and 1; noise_floor: what the rounding of logLike (a sum of N terms) alone puts into such a
    difference quotient.  Raises AssertionError when a residual exceeds rtol + noise_floor.  Works on any problem with
    sample_x_z / logLike_and_grad_z_logLike / grad_theta_logLike."""
    import numpy as np
    from .problem import SimRng
    theta = np.atleast_1d(np.asarray(theta, dtype=np.float64))
    x, z = prob.sample_x_z(SimRng(int(rng), 0), theta)
    z = 0.8 * np.asarray(z) + 0.05        # off the draw: residuals and latent values both non-zero
    f0, g = prob.logLike_and_grad_z_logLike(x, z, theta)
    N = z.size
    eps = np.finfo(np.float64).eps
    res_z = res_t = floor = 0.0
    for i in np.unique(np.linspace(0, N - 1, n_probe).astype(int)):
        e = np.zeros(N)
        e[i] = step * max(1.0, abs(z[i]))
        fp = prob.logLike_and_grad_z_logLike(x, z + e, theta)[0]
        fm = prob.logLike_and_grad_z_logLike(x, z - e, theta)[0]
        scale = max(1.0, abs(g[i]))
        floor = max(floor, 4 * eps * abs(f0) / e[i] / scale)
        res_z = max(res_z, abs((fp - fm) / (2 * e[i]) - g[i]) / scale)
    s = np.atleast_1d(prob.grad_theta_logLike(x, z, theta))
    for k in range(theta.size):
        d = np.zeros(theta.size)
        d[k] = step
        fp = prob.logLike_and_grad_z_logLike(x, z, theta + d)[0]
        fm = prob.logLike_and_grad_z_logLike(x, z, theta - d)[0]
        scale = max(1.0, abs(s[k]))
        floor = max(floor, 4 * eps * abs(f0) / step / scale)
        res_t = max(res_t, abs((fp - fm) / (2 * step) - s[k]) / scale)
    out = {"grad_z": float(res_z), "grad_theta": float(res_t), "noise_floor": float(floor)}
    if getattr(prob, "user_model", None) is not None and getattr(prob, "has_second_derivatives", False):
        out["second"] = _check_second(prob.model_eval, theta, np.asarray(x), z, n_probe, rtol)
    for name in ("grad_z", "grad_theta"):
        assert out[name] <= rtol + floor, (f"model consistency: {name} differs from the finite difference of logLike by {out[name]:.3g} "
                                           f"(relative; tolerance {rtol:g} + noise floor {floor:.3g})")
    return out


def _check_second(model_eval, theta, x, z, n_probe, rtol, h=1e-5):
    """The second-derivative functions of a header with MUSE_MODEL_SECOND (include/muse_model.h) against central differences of
    the header's own first-order functions, element by element on the host (model_eval: HipMuseProblem.model_eval): ozz = d grad / dz,
    ozx = d grad / dx, bz = dB / dz, bx = dB / dx, dx_dsd = d x(sd, n1, n2) / d sd.  Returns the worst residual (relative to the
    larger of the value's size and 1) and raises AssertionError beyond rtol."""
    import numpy as np
    N, nt = z.size, theta.size
    worst = 0.0
    rs = np.random.RandomState(7)
    for i in np.unique(np.linspace(0, N - 1, n_probe).astype(int)):
        k = int(i * nt // N)
        iv, sd = float(np.exp(-theta[k])), float(np.exp(0.5 * theta[k]))
        xi, zi = float(x[i]), float(z[i])
        n1, n2 = rs.randn(2)
        e = model_eval(iv, sd, xi, zi, n1, n2, i)
        hz, hx, hs = h * max(1.0, abs(zi)), h * max(1.0, abs(xi)), h * sd
        zp, zm = model_eval(iv, sd, xi, zi + hz, n1, n2, i), model_eval(iv, sd, xi, zi - hz, n1, n2, i)
        xp, xm = model_eval(iv, sd, xi + hx, zi, n1, n2, i), model_eval(iv, sd, xi - hx, zi, n1, n2, i)
        sp, sm = model_eval(iv, sd + hs, xi, zi, n1, n2, i), model_eval(iv, sd - hs, xi, zi, n1, n2, i)
        pairs = {"ozz": (zp["grad"] - zm["grad"]) / (2 * hz), "ozx": (xp["grad"] - xm["grad"]) / (2 * hx),
                 "bz": (zp["B"] - zm["B"]) / (2 * hz), "bx": (xp["B"] - xm["B"]) / (2 * hx),
                 "dx_dsd": (sp["x"] - sm["x"]) / (2 * hs)}
        for name, fd in pairs.items():
            res = abs(fd - e[name]) / max(1.0, abs(e[name]))
            assert res <= rtol, (f"model consistency: {name} of element {i} is {e[name]:.9g}, the finite difference of the header's "
                                 f"own functions gives {fd:.9g}")
            worst = max(worst, res)
    return float(worst)
