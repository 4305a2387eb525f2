"""L-BFGS (m = 10) with the Hager-Zhang line search on torch tensors: the DEFAULT `ẑ_at_θ` of the reference's problem interface
(src/interface.jl:140-166: `optimize(Optim.only_fg!(...), z₀, LBFGS(), Options(g_tol = ∇z_logLike_atol))`) for problems that bring
their own logLike as a closure (simple.TorchMuseProblem -- the reference's SimpleMuseProblem with AD, src/simple.jl:79-95).

NOT the hot path: the models on it (built-in, header-supplied, generated) are solved by the HIP kernels, one workgroup per simulation;
this is the general front-end behind them, for a logLike no header expresses -- every vector operation here is a torch operation on
the tensor's own device, and an evaluation costs what the closure's autograd costs.

Restated from the published algorithms as Optim.jl / LineSearches.jl configure them by default (the packages
are not vendored under the reference's tree): two-loop recursion with the gamma = s'y / y'y scaling of the initial Hessian, history
reset when 1/(dx'dg) is infinite, steepest descent when the direction is not a descent direction, `InitialStatic(alpha = 1)`,
HagerZhang (delta 0.1, sigma 0.9, rho 5, epsilon 1e-6, gamma 0.66, psi3 0.1, 50 iterations), convergence on |g|_inf <= g_tol, on an
unchanged point, or on an unchanged value twice in a row; at most 1000 iterations.  tests/test_simple_problem.py holds it against the CPU
checker's restatement of the same algorithms on the funnel objective: the same iteration and evaluation counts.
"""
import math

STATUS_G_CONVERGED, STATUS_X_CONVERGED, STATUS_F_CONVERGED, STATUS_MAXITER, STATUS_LINESEARCH_FAILED, STATUS_NONFINITE = range(6)

_DELTA, _SIGMA, _RHO, _EPSILON, _GAMMA, _PSI3, _LSMAX = 0.1, 0.9, 5.0, 1e-6, 0.66, 0.1, 50
_EPS = 2.220446049250313e-16


def _eps_of(b):
    return math.nextafter(abs(b), math.inf) - abs(b)


class _Objective:
    """value and gradient with the cache that decides f_calls: a point evaluated last is not evaluated again."""

    def __init__(self, fg):
        self.fg, self.x_last, self.f, self.g, self.f_calls = fg, None, math.nan, None, 0

    def at(self, x):
        import torch
        if self.x_last is None or not torch.equal(self.x_last, x):
            f, g = self.fg(x)
            self.f, self.g, self.x_last = float(f), g, x.clone()
            self.f_calls += 1
        return self.f, self.g


def _hagerzhang(obj, x, s, c, phi_0, dphi_0):
    """(alpha, ok): the step along s from x; ok False is LineSearches' exception (the caller stops)."""
    if not (math.isfinite(phi_0) and math.isfinite(dphi_0)) or dphi_0 >= _EPS * abs(phi_0):
        return 0.0, False
    if dphi_0 >= 0.0:
        return 0.0, True
    al, va, sl = [0.0], [phi_0], [dphi_0]          # the trace of trial points

    def phidphi(a):
        f, g = obj.at(x + a * s)
        return f, float(g.dot(s))

    def push(a, v, d):
        al.append(a); va.append(v); sl.append(d)
        return len(al) - 1

    fin = lambda v, d: math.isfinite(v) and math.isfinite(d)
    phi_lim = phi_0 + _EPSILON * abs(phi_0)

    def wolfe(cc, phi_c, dphi_c):
        w1 = _DELTA * dphi_0 >= (phi_c - phi_0) / cc and dphi_c >= _SIGMA * dphi_0
        w2 = (2.0 * _DELTA - 1.0) * dphi_0 >= dphi_c >= _SIGMA * dphi_0 and phi_c <= phi_lim
        return w1 or w2

    def bisect(ia, ib):                              # stage U3
        a, b = al[ia], al[ib]
        if not (sl[ia] < 0.0 and va[ia] <= phi_lim and sl[ib] < 0.0 and va[ib] > phi_lim and b > a):
            return None
        while b - a > _eps_of(b):
            d = (a + b) / 2.0
            v, g = phidphi(d)
            if not fin(v, g):
                return None
            i = push(d, v, g)
            if g >= 0.0:
                return ia, i
            if v <= phi_lim:
                a, ia = d, i
            else:
                b, ib = d, i
        return ia, ib

    def update(ia, ib, ic):                          # stages U0-U3
        a, b = al[ia], al[ib]
        if not (sl[ia] < 0.0 and va[ia] <= phi_lim and sl[ib] >= 0.0 and b > a):
            return None
        cc = al[ic]
        if cc < a or cc > b:
            return ia, ib
        if sl[ic] >= 0.0:
            return ia, ic
        if va[ic] <= phi_lim:
            return ic, ib
        return bisect(ia, ic)

    secant = lambda a, b, da, db: (a * db - b * da) / (db - da)

    def secant2(ia, ib):                             # stages S1-S4: (wolfe, iA, iB) or None
        a, b, da, db = al[ia], al[ib], sl[ia], sl[ib]
        if not (da < 0.0 and db >= 0.0):
            return None
        cc = secant(a, b, da, db)
        if not math.isfinite(cc):
            return None
        v, g = phidphi(cc)
        if not fin(v, g):
            return None
        ic = push(cc, v, g)
        if wolfe(cc, v, g):
            return True, ic, ic
        r = update(ia, ib, ic)
        if r is None:
            return None
        iA, iB = r
        a, b = al[iA], al[iB]
        if iB == ic:
            cc = secant(al[ib], al[iB], sl[ib], sl[iB])
        elif iA == ic:
            cc = secant(al[ia], al[iA], sl[ia], sl[iA])
        if (iA == ic or iB == ic) and a <= cc <= b:
            v, g = phidphi(cc)
            if not fin(v, g):
                return None
            ic = push(cc, v, g)
            if wolfe(cc, v, g):
                return True, ic, ic
            r = update(iA, iB, ic)
            if r is None:
                return None
            iA, iB = r
        return False, iA, iB

    if c <= _EPS:
        return 0.0, True
    alphamax = math.inf
    phi_c, dphi_c = phidphi(c)
    k = 1
    while not fin(phi_c, dphi_c) and k < 52:
        k += 1
        c *= _PSI3
        phi_c, dphi_c = phidphi(c)
    if not fin(phi_c, dphi_c):
        return 0.0, True
    push(c, phi_c, dphi_c)
    bracketed, ia, ib, it = False, 0, 1, 1
    while not bracketed and it < _LSMAX:
        if dphi_c >= 0.0:
            ib = len(al) - 1
            for i in range(ib - 1, -1, -1):
                if va[i] <= phi_lim:
                    ia = i
                    break
            bracketed = True
        elif va[-1] > phi_lim:
            r = bisect(0, len(al) - 1)
            if r is None:
                return al[0], False
            ia, ib = r
            bracketed = True
        else:
            cold = c
            if math.nextafter(cold, math.inf) >= alphamax:
                return cold, True
            c = min(c * _RHO, alphamax)
            phi_c, dphi_c = phidphi(c)
            k = 1
            while not fin(phi_c, dphi_c) and c > math.nextafter(cold, math.inf) and k < 52:
                alphamax = c
                k += 1
                c = (cold + c) / 2.0
                phi_c, dphi_c = phidphi(c)
            if not fin(phi_c, dphi_c):
                return cold, True
            push(c, phi_c, dphi_c)
        it += 1
    while it < _LSMAX:
        a, b = al[ia], al[ib]
        if not b > a:
            return a, False
        if b - a <= _eps_of(b):
            return a, True
        r = secant2(ia, ib)
        if r is None:
            return a, False
        w, iA, iB = r
        if w:
            return al[iA], True
        A, B = al[iA], al[iB]
        if not B > A:
            return A, False
        if B - A < _GAMMA * (b - a):
            if math.nextafter(va[ia], math.inf) >= va[ib] and math.nextafter(va[iA], math.inf) >= va[iB]:
                return A, True                        # flat: the secant steps did nothing useful
            ia, ib = iA, iB
        else:
            c = (A + B) / 2.0
            phi_c, dphi_c = phidphi(c)
            if not fin(phi_c, dphi_c):
                return A, False
            r = update(iA, iB, push(c, phi_c, dphi_c))
            if r is None:
                return A, False
            ia, ib = r
        it += 1
    return al[ia], False                              # linesearchmax reached


def lbfgs(fg, x0, g_tol, m=10, maxiter=1000):
    """Minimise f from x0 (a 1-d float64 torch tensor, any device).  fg(x) -> (f, g): the value (a float or 0-d tensor) and the
    gradient (a tensor like x).  Returns (x, info) with info = {iterations, f_calls, status, f_min, gnorm}."""
    import torch
    obj = _Objective(fg)
    x = x0.detach().clone()
    f, g = obj.at(x)
    gmax = lambda v: float(v.abs().max()) if v.numel() else 0.0
    iterations, pseudo, status, counter_f = 0, 0, STATUS_MAXITER, 0
    done = False
    if not math.isfinite(f) or not bool(torch.isfinite(g).all()):
        status, done = STATUS_NONFINITE, True
    elif gmax(g) <= g_tol:
        status, done = STATUS_G_CONVERGED, True
    dxs, dgs, rho = [None] * m, [None] * m, [0.0] * m
    while not done and iterations < maxiter:
        iterations += 1
        pseudo += 1
        lower, upper = pseudo - m, pseudo - 1
        q = g.clone()
        alpha_i = {}
        for index in range(upper, max(lower, 1) - 1, -1):           # twoloop!
            i = (index - 1) % m
            alpha_i[i] = rho[i] * float(dxs[i].dot(q))
            q -= alpha_i[i] * dgs[i]
        if pseudo > 1:
            i = (upper - 1) % m
            s = (float(dxs[i].dot(dgs[i])) / float(dgs[i].dot(dgs[i]))) * q
        else:
            s = q
        for index in range(max(lower, 1), upper + 1):
            i = (index - 1) % m
            s = s + dxs[i] * (alpha_i[i] - rho[i] * float(dgs[i].dot(s)))
        s = -s
        g_prev = g
        dphi_0 = float(g.dot(s))
        if dphi_0 >= 0.0:                                            # reset_search_direction!
            pseudo = 1
            s = -g
            dphi_0 = float(g.dot(s))
        f_prev, x_prev = f, x
        alpha, ok = _hagerzhang(obj, x, s, 1.0, f, dphi_0)
        dx = alpha * s
        x = x + dx
        if not ok:                       # (the value and gradient reported are those of the last point evaluated)
            status = STATUS_LINESEARCH_FAILED
            f, g = obj.f, obj.g
            break
        f, g = obj.at(x)
        x_conv = float((x - x_prev).abs().max()) <= 0.0 if x.numel() else True
        f_conv = abs(f - f_prev) <= 0.0
        g_conv = gmax(g) <= g_tol
        counter_f = counter_f + 1 if f_conv else 0
        done = x_conv or g_conv or counter_f > 1
        if g_conv:
            status = STATUS_G_CONVERGED
        elif x_conv:
            status = STATUS_X_CONVERGED
        elif counter_f > 1:
            status = STATUS_F_CONVERGED
        dg = g - g_prev
        denom = float(dx.dot(dg))
        rho_it = math.inf if denom == 0.0 else 1.0 / denom
        if math.isinf(rho_it):
            pseudo = 0
        else:
            i = (pseudo - 1) % m
            dxs[i], dgs[i], rho[i] = dx, dg, rho_it
        if not bool(torch.isfinite(g).all()):
            status = STATUS_NONFINITE
            break
    if status <= STATUS_F_CONVERGED and not math.isfinite(f):
        status = STATUS_NONFINITE
    return x, {"iterations": iterations, "f_calls": obj.f_calls, "status": status, "f_min": f, "gnorm": gmax(g)}
