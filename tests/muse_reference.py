"""An independent restatement of the host algebra of the reference's drivers -- TEST INFRASTRUCTURE.

Written from the Julia source alone (src/muse.jl:112-250 `muse!`, :484-532 `get_J!`, :407-446 `get_H!` FD
branch, :535-549 `finalize_result!`), in plain Python loops over lists of floats; it imports nothing from the
product package (museinference.jl_amd/) and shares no code with it or with the library's native `muse_run`.
The pmap body -- sample, MAP, scores -- is a callback: the tests plug in the CPU oracle or the HIP engine, so
that what is compared is exactly the algebra AFTER the map (means, corrected variances, the Broyden updates,
the posterior Hessian, the Newton-Raphson step, `regularize`, the convergence test, J, H, Sigma).

Matrices are lists of rows; inverses by Gauss-Jordan with partial pivoting (nθ <= 8).
"""
import math


# ---- small dense algebra on lists ---------------------------------------------------------------
def _zeros(n, m):
    return [[0.0] * m for _ in range(n)]


def _matvec(A, v):
    return [math.fsum(A[i][j] * v[j] for j in range(len(v))) for i in range(len(A))]


def _matmul(A, B):
    n, k, m = len(A), len(B), len(B[0])
    return [[math.fsum(A[i][t] * B[t][j] for t in range(k)) for j in range(m)] for i in range(n)]


def _transpose(A):
    return [list(r) for r in zip(*A)]


def _inv(A):
    n = len(A)
    M = [list(map(float, A[i])) + [1.0 if i == j else 0.0 for j in range(n)] for i in range(n)]
    for c in range(n):
        p = max(range(c, n), key=lambda r: abs(M[r][c]))
        if M[p][c] == 0.0:
            raise ZeroDivisionError("singular matrix")
        M[c], M[p] = M[p], M[c]
        d = M[c][c]
        M[c] = [v / d for v in M[c]]
        for r in range(n):
            if r != c and M[r][c] != 0.0:
                f = M[r][c]
                M[r] = [a - f * b for a, b in zip(M[r], M[c])]
    return [row[n:] for row in M]


def _diag(v):
    n = len(v)
    return [[v[i] if i == j else 0.0 for j in range(n)] for i in range(n)]


def mean_rows(rows):
    """mean of a vector of vectors (Statistics.mean, src/muse.jl:183,446)."""
    n = len(rows)
    return [math.fsum(r[k] for r in rows) / n for k in range(len(rows[0]))]


def var_rows(rows):
    """elementwise corrected variance of a vector of vectors (Statistics.var, src/muse.jl:188)."""
    n, m = len(rows), mean_rows(rows)
    return [math.fsum((r[k] - m[k]) ** 2 for r in rows) / (n - 1) for k in range(len(m))]


def cov_rows(rows):
    """SimpleCovariance(corrected=true) (src/muse.jl:495,529): (n-1)-normalised sample covariance."""
    n, m = len(rows), mean_rows(rows)
    nt = len(m)
    return [[math.fsum((r[a] - m[a]) * (r[b] - m[b]) for r in rows) / (n - 1) for b in range(nt)] for a in range(nt)]


# ---- muse! ----------------------------------------------------------------------------------------
def muse_loop(map_body, theta0, *, nsims, prior_grad_t, prior_hess_t, transform=None, inv_transform=None, maxsteps=50,
              theta_rtol=1e-1, alpha=0.7, regularize=None, Hinv_like0=None, Hinv_update="sims",
              broyden_memory=math.inf, history=None):
    """The loop of muse! (src/muse.jl:159-236).

    map_body(i, theta, theta_t) -> (g, g_t): the pmap of :169-176 -- lists of nsims+1 score vectors (data element
    first) in the untransformed and the transformed space.  prior_grad_t / prior_hess_t: gradient and Hessian of
    logPriorθ(θ′, Transformedθ()) (the reference takes them by ForwardDiff, :184,207).
    Returns (history, theta_result, gs_result): history records hold the reference's field names; theta_result is
    result.θ (= θunreg, :230) and gs_result result.gs (:231).
    """
    ident = lambda t: list(t)
    transform = transform or ident
    inv_transform = inv_transform or ident
    regularize = regularize or ident
    alpha_fn = alpha if callable(alpha) else (lambda i: alpha)
    history = [] if history is None else history
    theta = theta_unreg = [float(t) for t in theta0]                       # :135
    theta_t = theta_unreg_t = list(transform(theta))                       # :136
    Hinv_like = Hinv_like0
    result_theta, result_gs = None, None
    nt = len(theta)
    i = len(history)
    while i < maxsteps:                                                    # for i = length(history)+1 : maxsteps
        i += 1
        if i > 2:                                                          # :163-166
            d = [a - b for a, b in zip(history[-1]["θ′"], history[-2]["θ′"])]
            q = -math.fsum(d[a] * v for a, v in enumerate(_matvec(history[-1]["H⁻¹_post′"], d)))
            if math.sqrt(q) < theta_rtol:                                  # raises ValueError for q < 0 (DomainError)
                break
        g, g_t = map_body(i, theta, theta_t)                               # :169-176
        g_dat_t, g_sims, g_sims_t = list(g_t[0]), [list(r) for r in g[1:]], [list(r) for r in g_t[1:]]
        m = mean_rows(g_sims_t)
        g_like_t = [a - b for a, b in zip(g_dat_t, m)]                     # :183
        g_prior_t = list(prior_grad_t(theta_t))                            # :184
        g_post_t = [a + b for a, b in zip(g_like_t, g_prior_t)]            # :185
        Hinv_like_sims = _diag([-1.0 / v for v in var_rows(g_sims_t)])     # :188-189
        if Hinv_like is None or Hinv_update == "sims":                     # :190-191
            Hinv_like = Hinv_like_sims
        elif i > 2 and Hinv_update in ("broyden", "diagonal_broyden"):     # :192-205
            j0 = int(max(2, i - broyden_memory))
            Hinv_like = history[j0 - 2]["H⁻¹_like_sims′"]                  # history[j₀-1], 1-based
            for j in range(j0, i):                                         # j = j₀ : i-1
                dth = [a - b for a, b in zip(history[j - 1]["θ′"], history[j - 2]["θ′"])]
                dg = [a - b for a, b in zip(history[j - 1]["g_like′"], history[j - 2]["g_like′"])]
                Hdg = _matvec(Hinv_like, dg)
                denom = math.fsum(dth[a] * Hdg[a] for a in range(nt))      # Δθ′' * H⁻¹ * Δg
                u = [(dth[a] - Hdg[a]) / denom for a in range(nt)]
                row = _matvec(_transpose(Hinv_like), dth)                  # Δθ′' * H⁻¹  (a row vector)
                Hinv_like = [[Hinv_like[a][b] + u[a] * row[b] for b in range(nt)] for a in range(nt)]
                if Hinv_update == "diagonal_broyden":
                    Hinv_like = _diag([Hinv_like[a][a] for a in range(nt)])
        H_prior_t = [list(r) for r in prior_hess_t(theta_t)]               # :207
        inner = _inv(Hinv_like)
        Hinv_post = _inv([[inner[a][b] + H_prior_t[a][b] for b in range(nt)] for a in range(nt)])   # :208
        history.append({"θ": theta, "θunreg": theta_unreg, "θ′": theta_t, "θunreg′": theta_unreg_t,
                        "g_like_sims": g_sims, "g_like_dat′": g_dat_t, "g_like_sims′": g_sims_t, "g_like′": g_like_t,
                        "g_prior′": g_prior_t, "g_post′": g_post_t, "H⁻¹_post′": Hinv_post, "H_prior′": H_prior_t,
                        "H⁻¹_like′": Hinv_like, "H⁻¹_like_sims′": Hinv_like_sims})   # :211-221
        step = _matvec(Hinv_post, g_post_t)
        a_i = alpha_fn(i)
        theta_unreg_t = [t - a_i * s for t, s in zip(theta_t, step)]       # :224
        theta_unreg = list(inv_transform(theta_unreg_t))                   # :225
        theta_t = list(regularize(theta_unreg_t))                          # :226
        theta = list(inv_transform(theta_t))                               # :227
        result_theta, result_gs = theta_unreg, g_sims                      # :230-231
    return history, result_theta, result_gs


# ---- get_J!, get_H!, finalize_result! ---------------------------------------------------------------
def J_from_scores(gs):
    """J = cov(SimpleCovariance(corrected=true), gs) (src/muse.jl:529); var for scalar θ is its 1x1 case."""
    return cov_rows([list(g) for g in gs])


def fd_step_from_scores(gs):
    """step = 0.1 ./ std(result.gs) (src/muse.jl:411-413)."""
    return [0.1 / math.sqrt(v) for v in var_rows([list(g) for g in gs])]


def central_fdm_3_1(f_plus, f_minus, h):
    """central_fdm(3,1) with an explicit step (src/muse.jl:300, src/util.jl:13): grid (-1,0,1), coefficients
    (-1/2, 0, 1/2), divided by the step."""
    return [(-0.5 * m + 0.5 * p) / h for p, m in zip(f_plus, f_minus)]


def H_from_columns(cols_per_sim):
    """Per-sim Jacobian = hcat of the columns (src/util.jl:25): cols_per_sim[s][j][i] = d g_i / d θ_j;
    H = mean(Hs) (src/muse.jl:446).  Returns (Hs, H) with Hs[s][i][j]."""
    Hs = [_transpose(cols) for cols in cols_per_sim]
    n, nt = len(Hs), len(Hs[0])
    H = [[math.fsum(Hs[s][a][b] for s in range(n)) / n for b in range(nt)] for a in range(nt)]
    return Hs, H


def finalize(H, J, prior_hess_u):
    """finalize_result! (src/muse.jl:535-549): Σ⁻¹ = H' J⁻¹ H + H_prior, H_prior = -∇²θ logPrior(θ̂); Σ = inv(Σ⁻¹)."""
    nt = len(H)
    S_inv = _matmul(_matmul(_transpose(H), _inv(J)), H)
    S_inv = [[S_inv[a][b] - prior_hess_u[a][b] for b in range(nt)] for a in range(nt)]
    return S_inv, _inv(S_inv)


# ---- FiniteDifferences.jl: central_fdm(p, q) and the estimated step (reference: fdm at src/muse.jl:300, applied at
#      src/util.jl:13 as fdm(f, 0[, step])).  The package is not vendored in the reference (Project.toml:40, compat
#      0.12.20) and cannot run here: restated from its published algorithm AS RECALLED -- coefficients from the
#      Vandermonde system solved in exact rationals; step = argmin of  C1 h^-q + C2 h^(p-q),  C1 = eps(|f|) * sum|c| * factor,
#      C2 = |f^(p)| * sum|c g^p| / p!,  |f^(p)| and |f| from one evaluation of central_fdm(p + 2, p) at ITS default step
#      (|f^(p)| taken as `condition` = 10, error eps(Float64)), maxima over three neighbourhood estimates (the same values
#      with the grid shifted by -1, 0, +1); steps capped at 1000 x the default step. ------------------------------------
def fdm_central_grid(p):
    half = p // 2
    return list(range(-half, half + 1)) if p % 2 else [g for g in range(-half, half + 1) if g != 0]


def fdm_coefs(grid, q):
    from fractions import Fraction
    p = len(grid)
    rows = [[Fraction(g) ** i for g in grid] for i in range(p)]
    rhs = [Fraction(math.factorial(q) if i == q else 0) for i in range(p)]
    for c in range(p):                                   # Gauss-Jordan on exact rationals
        r = c
        while rows[r][c] == 0:
            r += 1
        rows[c], rows[r], rhs[c], rhs[r] = rows[r], rows[c], rhs[r], rhs[c]
        piv = rows[c][c]
        rows[c] = [v / piv for v in rows[c]]
        rhs[c] = rhs[c] / piv
        for k in range(p):
            if k != c and rows[k][c] != 0:
                fac = rows[k][c]
                rows[k] = [a - fac * b for a, b in zip(rows[k], rows[c])]
                rhs[k] = rhs[k] - fac * rhs[c]
    return [float(v) for v in rhs]


def _fdm_mults(grid, q):
    c = fdm_coefs(grid, q)
    p = len(grid)
    return c, sum(abs(ci * g ** p) for ci, g in zip(c, grid)) / math.factorial(p), sum(abs(ci) for ci in c)


def _fdm_step(p, q, grad_mult, err_mult, grad_magnitude, f_error, factor=1.0):
    c1 = f_error * err_mult * factor
    c2 = grad_magnitude * grad_mult
    return (q / (p - q) * (c1 / c2)) ** (1.0 / p)


def fdm_default_step(p, q, condition=10.0):
    _, gm, em = _fdm_mults(fdm_central_grid(p), q)
    return _fdm_step(p, q, gm, em, condition, 2.220446049250313e-16)


def fdm_estimate_step(f, p, q, x=0.0):
    """Step of central_fdm(p, q; adapt = 1) for f (float -> list of floats) at x."""
    pe, qe = p + 2, p                                    # the bound estimator: central_fdm(p + 2, p), not adapted
    ge = fdm_central_grid(pe)
    he = min(fdm_default_step(pe, qe), 1000.0 * fdm_default_step(pe, qe))
    vals = [list(f(x + he * g)) for g in ge]
    grad_mag = 0.0
    for shift in (-1, 0, 1):
        cs = fdm_coefs([g + shift for g in ge], qe)
        for k in range(len(vals[0])):
            grad_mag = max(grad_mag, abs(math.fsum(c * v[k] for c, v in zip(cs, vals)) / he ** qe))
    f_mag = max(abs(v) for row in vals for v in row)
    _, gm, em = _fdm_mults(fdm_central_grid(p), q)
    if grad_mag == 0.0 or f_mag == 0.0:
        h = fdm_default_step(p, q)
    else:
        h = _fdm_step(p, q, gm, em, grad_mag, math.ulp(f_mag))
    return min(h, 1000.0 * fdm_default_step(p, q))


def fdm_apply(f, p, q, x=0.0, step=None):
    """fdm(f, x[, step]) for central_fdm(p, q): sum_g c_g f(x + h g) / h^q, every grid point evaluated."""
    grid = fdm_central_grid(p)
    h = fdm_estimate_step(f, p, q, x) if step is None else step
    c = fdm_coefs(grid, q)
    vals = [list(f(x + h * g)) for g in grid]
    return [math.fsum(ci * v[k] for ci, v in zip(c, vals)) / h ** q for k in range(len(vals[0]))]
