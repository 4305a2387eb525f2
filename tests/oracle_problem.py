"""An AbstractMuseProblem whose operators are the CPU oracle -- TEST INFRASTRUCTURE.

Lets the host drivers (muse_, get_J_, get_H_) and the multi-process sharding be exercised without a
GPU, and serves as the checker the HIP-backed problem is compared with.  Never shipped: the product
package does not import it.
"""
import numpy as np

import museinference_jl_amd as M
from oracle import oracle as O


class OracleMuseProblem(M.AbstractMuseProblem):
    def __init__(self, x, model="funnel", ntheta=1, prior=None, N=None, batched=True, nthreads=4):
        from museinference_jl_amd.priors import as_prior
        self.x = None if x is None else np.asarray(x, dtype=np.float64)
        self.N = int(N) if x is None else self.x.size
        self.model, self.ntheta, self.prior = model, ntheta, as_prior(prior)
        self.nthreads = nthreads
        self._zhat = {}
        if not batched:  # hide the batched seams: drives the element-by-element path of the host drivers
            self.map_and_score_batch = None
            del self.map_and_score_batch

    def logPrior_theta(self, theta, theta_space=None):
        return self.prior.logpdf(np.asarray(theta, dtype=np.float64))

    def grad_logPrior_theta(self, theta, theta_space=None):
        return np.atleast_1d(self.prior.grad(np.asarray(theta, dtype=np.float64)))

    def hess_logPrior_theta(self, theta, theta_space=None):
        return np.atleast_2d(self.prior.hess(np.asarray(theta, dtype=np.float64)))

    def sample_x_z(self, rng, theta):
        return O.sample_x_z(self.model, self.N, rng.seed, rng.sim, theta)

    def logLike_and_grad_z_logLike(self, x, z, theta):
        return O.logLike_and_grad_z(self.model, x, z, theta)

    def grad_theta_logLike(self, x, z, theta, theta_space=None):
        return O.grad_theta(self.model, x, z, theta)

    def zhat_at_theta(self, x, z0, theta, grad_z_logLike_atol=1e-2):
        z, info = O.zhat_at_theta(self.model, x, z0, theta, grad_z_logLike_atol)
        rec = np.zeros(1, dtype=M._capi.INFO_DTYPE)
        for k in info:
            rec[k] = info[k]
        return z, rec[0]


class OracleBatchedProblem(OracleMuseProblem):
    """Adds the batched seams (same signatures as HipMuseProblem) on top of the oracle."""

    def map_and_score_batch(self, rng, sim_begin, sim_end, theta, *, include_data=False, atol=1e-2, z0_mode=0):
        seed = rng.seed if isinstance(rng, M.SimRng) else int(rng)
        n = (sim_end - sim_begin) + (1 if include_data else 0)
        zhat = None
        if z0_mode == M.Z0_WARM:
            zhat = np.stack([self._zhat.get(e, np.zeros(self.N)) for e in range(n)])
        g, zh, info = O.map_and_score_batch(self.model, self.N, seed, sim_begin, sim_end, theta, atol=atol,
                                            x_data=self.x if include_data else None, z0_mode=z0_mode, zhat=zhat,
                                            nthreads=self.nthreads)
        for e in range(n):
            self._zhat[e] = zh[e]
        return g, info.astype(M._capi.INFO_DTYPE)

    def get_zhat(self, b, e):
        return np.stack([self._zhat[k] for k in range(b, e)])

    def set_zhat(self, b, zs):
        for k, z in enumerate(np.atleast_2d(zs)):
            self._zhat[b + k] = np.array(z, dtype=np.float64)

    def fd_jacobian_batch(self, rng, sim_begin, sim_end, theta0, step, *, atol=1e-2, fid_mode=0, fid_sim=M.MASTER_SIM):
        seed = rng.seed if isinstance(rng, M.SimRng) else int(rng)
        th = np.atleast_1d(np.asarray(theta0, dtype=np.float64))
        Hs = []
        for s in range(sim_begin, sim_end):
            fid = fid_sim if fid_mode == 0 else s
            _, zfid, _ = O.map_and_score_batch(self.model, self.N, seed, fid, fid + 1, th, atol=atol, z0_mode=0)
            Hs.append(O.fd_jacobian(self.model, self.N, seed, s, th, step, zfid[0], atol=atol))
        info = np.zeros((sim_end - sim_begin, th.size, 2), dtype=M._capi.INFO_DTYPE)
        return np.array(Hs).reshape(sim_end - sim_begin, th.size, th.size), info

    def fd_jacobian_columns(self, rng, sim_begin, col_begin, col_end, theta0, step, *, atol=1e-2, fid_mode=0,
                            fid_sim=M.MASTER_SIM):
        nth = np.atleast_1d(theta0).size
        s_lo, s_hi = sim_begin + col_begin // nth, sim_begin + (col_end - 1) // nth + 1
        Hs, _ = self.fd_jacobian_batch(rng, s_lo, s_hi, theta0, step, atol=atol, fid_mode=fid_mode, fid_sim=fid_sim)
        self.fd_maps_done = getattr(self, "fd_maps_done", 0) + 2 * (col_end - col_begin)
        cols = Hs.transpose(0, 2, 1).reshape(-1, nth)[col_begin - (s_lo - sim_begin) * nth:][: col_end - col_begin]
        return np.ascontiguousarray(cols), np.zeros((col_end - col_begin, 2), dtype=M._capi.INFO_DTYPE)

    def fd_values_columns(self, rng, sim_begin, col_begin, col_end, theta0, offsets, *, per_unit=False, atol=1e-2, fid_mode=0,
                          fid_sim=M.MASTER_SIM):
        """The raw finite-difference values on the oracle, unit by unit: score at theta0 of the simulation drawn at
        theta0 + offset e_j, MAP at theta0 from the fiducial MAP (src/muse.jl:426-432)."""
        seed = rng.seed if isinstance(rng, M.SimRng) else int(rng)
        th = np.atleast_1d(np.asarray(theta0, dtype=np.float64))
        nth, off = th.size, np.asarray(offsets, dtype=np.float64)
        n, G = col_end - col_begin, off.shape[1]
        F = np.empty((n, G, nth))
        info = np.zeros((n, G), dtype=M._capi.INFO_DTYPE)
        zfids = {}
        for u in range(n):
            e = col_begin + u
            s, j = sim_begin + e // nth, e % nth
            fid = fid_sim if fid_mode == 0 else s
            if fid not in zfids:
                zfids[fid] = O.map_and_score_batch(self.model, self.N, seed, fid, fid + 1, th, atol=atol, z0_mode=0)[1][0]
            for g in range(G):
                t = th.copy()
                t[j] = th[j] + off[u if per_unit else j, g]
                x, _ = O.sample_x_z(self.model, self.N, seed, s, t)
                zh, inf = O.zhat_at_theta(self.model, x, zfids[fid], th, atol)
                F[u, g] = O.grad_theta(self.model, x, zh, th)
                for k in inf:
                    info[u, g][k] = inf[k]
        self.fd_maps_done = getattr(self, "fd_maps_done", 0) + n * G
        return F, info

    def implicit_H_columns(self, rng, sim_begin, col_begin, col_end, theta0, *, atol=1e-1, cg_maxiter=100):
        nth = np.atleast_1d(theta0).size
        s_lo, s_hi = sim_begin + col_begin // nth, sim_begin + (col_end - 1) // nth + 1
        Hs, its = self.implicit_H_batch(rng, s_lo, s_hi, theta0, atol=atol, cg_maxiter=cg_maxiter)
        off = col_begin - (s_lo - sim_begin) * nth
        cols = Hs.transpose(0, 2, 1).reshape(-1, nth)[off:][: col_end - col_begin]
        return np.ascontiguousarray(cols), np.asarray(its).reshape(-1)[off:][: col_end - col_begin].astype(np.int32)

    def implicit_H_batch(self, rng, sim_begin, sim_end, theta0, *, atol=1e-1, cg_maxiter=100):
        seed = rng.seed if isinstance(rng, M.SimRng) else int(rng)
        out = [O.implicit_H(self.model, self.N, seed, s, theta0, atol, cg_maxiter) for s in range(sim_begin, sim_end)]
        return np.array([h for h, _ in out]), np.array([i for _, i in out])
