"""Sharding of the sims of one map over the GPUs of a node, one process per GPU.

The reference parallelises the same map with Distributed.pmap over a worker pool (src/util.jl:74-83,
src/muse.jl:169,417,426,508).  Here rank r owns a contiguous block of the sim indices (its ẑ warm-start
state stays resident on its GPU), the data element lives on rank 0, and the only exchange is ONE
all-gather of the per-rank score blocks per map (<= 64 KB; latency-bound, never xGMI-bandwidth-bound),
after which every rank holds all scores in the reference's sim order and does the same host algebra --
so a result is bitwise independent of the number of GPUs.  Collectives go through torch.distributed
(backend "nccl" = RCCL over xGMI on GPUs; "gloo" on CPU for the multi-process tests).
"""
import os
import sys
import time

import numpy as np

from . import _capi


def ranks_share_node(dist, group=None):
    """True when every rank of the group runs under ONE kernel and sees ONE /dev/shm, which is what the engine's
    shared-memory transport needs.  Host names do not say that: containers on different nodes often share one, and
    containers on one node may have private /dev/shm mounts.  Rank 0 writes its boot id
    (/proc/sys/kernel/random/boot_id) into a probe file under /dev/shm; a rank agrees when it finds the file and
    its own boot id in it.  Collective over the group: the same answer on every rank."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    src = dist.get_global_rank(group, 0) if group is not None else 0
    try:
        boot = open("/proc/sys/kernel/random/boot_id").read().strip()
    except OSError:
        boot = ""
    token = [None]
    if rank == 0 and boot:
        name = f"/dev/shm/muse_probe_{os.getpid()}_{time.time_ns():x}"
        try:
            fd = os.open(name, os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o600)
            os.write(fd, boot.encode())
            os.close(fd)
            token[0] = name
        except OSError:
            token[0] = None
    dist.broadcast_object_list(token, src=src, group=group)
    ok = False
    if token[0] is not None and boot:
        try:
            ok = open(token[0]).read().strip() == boot
        except OSError:
            ok = False
    oks = [None] * world
    dist.all_gather_object(oks, bool(ok), group=group)  # also the barrier before the probe file goes away
    if rank == 0 and token[0] is not None:
        try:
            os.unlink(token[0])
        except OSError:
            pass
    return all(oks)


def block_partition(begin, end, world, rank):
    """Contiguous static partition of [begin, end): the first (n mod world) ranks get one extra."""
    n = end - begin
    base, extra = divmod(n, world)
    lo = begin + rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedMuseProblem:
    """Wraps a local problem (with batched seams) so that muse_/get_J_/get_H_ run sharded.

    All attribute access other than the batched seams is forwarded to the local problem."""

    def __init__(self, local, group=None, device=None, engine_comm=None, transport=None):
        """engine_comm: exchange through the engine's own communicator (muse_comm_* of the C ABI, no torch tensors on
        the path).  transport "shm": the ranks share a node and exchange their blocks host to host through a
        shared-memory segment; "rccl": pinned host -> device -> ncclAllGather -> pinned host.  Defaults, when the local
        problem is a HipMuseProblem: "shm" if the ranks share a node (ranks_share_node: one boot id, one /dev/shm), else
        "rccl" if the process group's backend is nccl (= RCCL); a transport that fails to come up on ANY rank is given
        up by ALL ranks together and the next one is tried; torch.distributed collectives otherwise (gloo on CPU).
        The unique id travels over the process group once.  `transport` reports what is in use (None: torch)."""
        import torch.distributed as dist
        self._dist = dist
        self.local = local
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._device = device
        self._last_nslots = None
        self._last_had_data = False
        has_engine = hasattr(local, "comm_init")
        nccl = dist.get_backend(group) == "nccl"
        self.engine_comm = False
        self.transport = None
        if not has_engine or engine_comm is False:
            return
        if getattr(local, "_nranks", None) is not None:  # the caller initialised the local communicator itself
            self.engine_comm = True
            self.transport = local.comm_transport()
            return
        # Transports to try, in order; an explicit request is tried alone (then torch.distributed).  Whether one works
        # is a COLLECTIVE decision: every rank reports, and all ranks move on to the next candidate together when any
        # of them failed -- a rank must never sit in a segment's or a communicator's time-out alone.
        if transport is not None:
            candidates = [transport]
        else:
            candidates = (["shm"] if ranks_share_node(dist, group) else []) + (["rccl"] if nccl else [])
        if engine_comm is None and not candidates:
            return
        src = dist.get_global_rank(group, 0) if group is not None else 0
        for cand in candidates:
            ok, why = True, ""
            uid = [None]
            if self.rank == 0:
                try:
                    uid[0] = type(local).comm_unique_id(cand)
                except Exception as e:  # noqa: BLE001
                    why = str(e)
            dist.broadcast_object_list(uid, src=src, group=group)
            if uid[0] is None:
                ok = False
            else:
                try:
                    local.comm_init(self.world, self.rank, uid[0])
                except Exception as e:  # noqa: BLE001
                    ok, why = False, str(e)
            oks = [None] * self.world
            dist.all_gather_object(oks, ok, group=group)
            if all(oks):
                self.engine_comm = True
                self.transport = cand
                return
            if ok:  # this rank was fine, a peer was not: give the communicator back
                local.comm_destroy()
            if self.rank == 0:
                print(f"[museinference] engine communicator '{cand}' unavailable on {oks.count(False)} rank(s)"
                      + (f" ({why})" if why else "") + "; trying the next transport", file=sys.stderr)
        if engine_comm is True and transport is not None:
            raise RuntimeError(f"engine communicator '{transport}' could not be initialised on every rank")

    def __getattr__(self, name):
        return getattr(self.local, name)

    # -- the muse! outer loop in the library's native code, sharded (muse_run_sharded of the C ABI): with the engine's own
    #    communicator every rank runs the loop itself -- one gathered map per iteration, the same step on every rank --
    #    and no Python, torch tensor or allocation sits between two maps.  Without it muse_() drives the maps from Python.
    supports_native_muse = True

    def native_prior(self):
        return self.local.native_prior() if self.engine_comm and hasattr(self.local, "native_prior") else None

    def run_muse(self, rng, theta0, *, nsims, maxsteps, theta_rtol, atol, alpha, z0_warm=False, device_loop=None):
        """As HipMuseProblem.run_muse, over the ranks: (n, theta, hist [n, W], g_sims [n, nsims, nθ], info [n, nsims+1]) --
        the same on every rank, and the same bits as the unsharded loop's."""
        loc = self.local
        lo, hi = block_partition(0, nsims, self.world, self.rank)
        nloc = (hi - lo) + (1 if self.rank == 0 else 0)
        n, theta, hist, gs, info = loc.run_muse_sharded(rng, theta0, nsims=nsims, maxsteps=maxsteps, theta_rtol=theta_rtol, atol=atol,
                                                        alpha=alpha, z0_warm=z0_warm)
        self._last_nslots, self._last_had_data = nsims + 1, True
        # the solver infos of every element, once for the run: [n iterations][this rank's elements] -> [n][nsims + 1]
        ninfo = len(_capi.INFO_DTYPE.names)
        counts = [(h - l) + (1 if r == 0 else 0) for r, (l, h) in
                  ((r, block_partition(0, nsims, self.world, r)) for r in range(self.world))]
        rows = self._info_to_rows(info[:n].T.reshape(-1)).reshape(nloc, n * ninfo) if n else np.zeros((nloc, 0))
        allrows = self._allgather_rows(rows, counts) if n else np.zeros((nsims + 1, 0))
        info_all = np.zeros((n, nsims + 1), dtype=_capi.INFO_DTYPE)
        for e in range(nsims + 1 if n else 0):
            info_all[:, e] = self._rows_to_info(allrows[e].reshape(n, ninfo))
        return n, theta, hist, gs, info_all

    def _tensor_device(self):
        import torch
        if self._device is not None:
            return self._device
        return torch.device("cuda", torch.cuda.current_device()) if self._dist.get_backend(self.group) == "nccl" \
            else torch.device("cpu")

    def _allgather_rows(self, rows, counts):
        """rows: [count_r, width] float64 on this rank -> concatenation over ranks in rank order."""
        import torch
        dev = self._tensor_device()
        width = rows.shape[1] if rows.ndim == 2 else 0
        cmax = max(counts)
        send = np.zeros((cmax, width))
        send[: rows.shape[0]] = rows
        if self.engine_comm:
            recv = self.local.allgather_scores(send.reshape(-1)).reshape(self.world, cmax, width)
            return np.concatenate([recv[r, :c] for r, c in enumerate(counts)], axis=0)
        t = torch.from_numpy(send).to(dev)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(out, t, group=self.group)
        return np.concatenate([o.cpu().numpy()[:c] for o, c in zip(out, counts)], axis=0)

    @staticmethod
    def _info_to_rows(info):
        flat = np.asarray(info).reshape(-1)
        return np.stack([flat[n].astype(np.float64) for n in _capi.INFO_DTYPE.names], axis=1)

    @staticmethod
    def _rows_to_info(rows):
        info = np.zeros(rows.shape[0], dtype=_capi.INFO_DTYPE)
        for k, n in enumerate(_capi.INFO_DTYPE.names):
            info[n] = rows[:, k]
        return info

    def map_and_score_batch(self, rng, sim_begin, sim_end, theta, *, include_data=False, atol=1e-2,
                            z0_mode=_capi.Z0_ZERO):
        lo, hi = block_partition(sim_begin, sim_end, self.world, self.rank)
        data_here = include_data and self.rank == 0
        self._last_nslots = (sim_end - sim_begin) + (1 if include_data else 0)
        self._last_had_data = bool(include_data)
        g, info = self.local.map_and_score_batch(rng, lo, hi, theta, include_data=data_here, atol=atol,
                                                 z0_mode=z0_mode)
        counts = []
        for r in range(self.world):
            l, h = block_partition(sim_begin, sim_end, self.world, r)
            counts.append((h - l) + (1 if include_data and r == 0 else 0))
        nth = g.shape[1]
        rows = np.concatenate([g, self._info_to_rows(info)], axis=1)
        allrows = self._allgather_rows(rows, counts)
        return np.ascontiguousarray(allrows[:, :nth]), self._rows_to_info(allrows[:, nth:])

    def fd_jacobian_batch(self, rng, sim_begin, sim_end, theta0, step, *, atol=1e-2, fid_mode=0, fid_sim=None):
        """get_H!'s finite-difference map.  The reference parallelises over whichever of sims / Jacobian columns is
        longer (src/muse.jl:327-333); here the flattened list of (sim, column) units is cut into contiguous blocks,
        one per rank, so that nsims = 3, nθ = 4 on two ranks is 6 columns (12 MAPs) each -- a block may begin and end
        inside a simulation's Jacobian.  Problems without the column seam are sharded by sims."""
        from .problem import MASTER_SIM
        fid_sim = MASTER_SIM if fid_sim is None else fid_sim
        nth = np.atleast_1d(theta0).size
        ninfo = len(_capi.INFO_DTYPE.names)
        if hasattr(self.local, "fd_jacobian_columns"):
            ncol = (sim_end - sim_begin) * nth
            lo, hi = block_partition(0, ncol, self.world, self.rank)
            if hi > lo:
                cols, info = self.local.fd_jacobian_columns(rng, sim_begin, lo, hi, theta0, step, atol=atol,
                                                            fid_mode=fid_mode, fid_sim=fid_sim)
            else:
                cols, info = np.zeros((0, nth)), np.zeros((0, 2), dtype=_capi.INFO_DTYPE)
            counts = [h - l for l, h in (block_partition(0, ncol, self.world, r) for r in range(self.world))]
            rows = np.concatenate([cols, self._info_to_rows(info).reshape(hi - lo, 2 * ninfo)], axis=1)
            allrows = self._allgather_rows(rows, counts)
            n = sim_end - sim_begin
            # the per-sim Jacobian is the hcat of its columns (src/util.jl:25): Hs[s][i][j] = cols[s*nθ + j][i]
            Hall = allrows[:, :nth].reshape(n, nth, nth).transpose(0, 2, 1)
            iall = self._rows_to_info(allrows[:, nth:].reshape(n * nth * 2, ninfo)).reshape(n, nth, 2)
            return np.ascontiguousarray(Hall), iall
        lo, hi = block_partition(sim_begin, sim_end, self.world, self.rank)
        if hi > lo:
            Hs, info = self.local.fd_jacobian_batch(rng, lo, hi, theta0, step, atol=atol, fid_mode=fid_mode,
                                                    fid_sim=fid_sim)
        else:
            Hs, info = np.zeros((0, nth, nth)), np.zeros((0, nth, 2), dtype=_capi.INFO_DTYPE)
        counts = [block_partition(sim_begin, sim_end, self.world, r) for r in range(self.world)]
        counts = [h - l for l, h in counts]
        rows = np.concatenate([Hs.reshape(hi - lo, nth * nth),
                               self._info_to_rows(info).reshape(hi - lo, nth * 2 * ninfo)], axis=1)
        allrows = self._allgather_rows(rows, counts)
        n = allrows.shape[0]
        Hall = allrows[:, : nth * nth].reshape(n, nth, nth)
        iall = self._rows_to_info(allrows[:, nth * nth:].reshape(n * nth * 2, ninfo)).reshape(n, nth, 2)
        return np.ascontiguousarray(Hall), iall

    def fd_values_columns(self, rng, sim_begin, col_begin, col_end, theta0, offsets, *, per_unit=False, atol=1e-2, fid_mode=0,
                          fid_sim=None):
        """The raw finite-difference values (any central_fdm(p, 1), estimated steps): the (sim, column) units are cut into
        contiguous blocks, one per rank, every rank ends up with all of them.  (F [n, G, nθ], info [n, G])"""
        from .problem import MASTER_SIM
        fid_sim = MASTER_SIM if fid_sim is None else fid_sim
        nth = np.atleast_1d(theta0).size
        ninfo = len(_capi.INFO_DTYPE.names)
        off = np.asarray(offsets, dtype=np.float64)
        G = off.shape[1]
        lo, hi = block_partition(col_begin, col_end, self.world, self.rank)
        if hi > lo:
            F, info = self.local.fd_values_columns(rng, sim_begin, lo, hi, theta0, off[lo - col_begin:hi - col_begin] if per_unit else off,
                                                   per_unit=per_unit, atol=atol, fid_mode=fid_mode, fid_sim=fid_sim)
        else:
            F, info = np.zeros((0, G, nth)), np.zeros((0, G), dtype=_capi.INFO_DTYPE)
        counts = [h - l for l, h in (block_partition(col_begin, col_end, self.world, r) for r in range(self.world))]
        rows = np.concatenate([F.reshape(hi - lo, G * nth), self._info_to_rows(info).reshape(hi - lo, G * ninfo)], axis=1)
        allrows = self._allgather_rows(rows, counts)
        n = col_end - col_begin
        return (np.ascontiguousarray(allrows[:, :G * nth].reshape(n, G, nth)),
                self._rows_to_info(allrows[:, G * nth:].reshape(n * G, ninfo)).reshape(n, G))

    def implicit_H_batch(self, rng, sim_begin, sim_end, theta0, *, atol=1e-1, cg_maxiter=100):
        """get_H! implicit-differentiation branch (src/muse.jl:335-405), the (sim, column) units shared like the
        finite-difference ones."""
        nth = np.atleast_1d(theta0).size
        if hasattr(self.local, "implicit_H_columns"):
            ncol = (sim_end - sim_begin) * nth
            lo, hi = block_partition(0, ncol, self.world, self.rank)
            if hi > lo:
                cols, its = self.local.implicit_H_columns(rng, sim_begin, lo, hi, theta0, atol=atol, cg_maxiter=cg_maxiter)
            else:
                cols, its = np.zeros((0, nth)), np.zeros(0, dtype=np.int32)
            counts = [h - l for l, h in (block_partition(0, ncol, self.world, r) for r in range(self.world))]
            rows = np.concatenate([cols, its.astype(np.float64).reshape(hi - lo, 1)], axis=1)
            allrows = self._allgather_rows(rows, counts)
            n = sim_end - sim_begin
            return (np.ascontiguousarray(allrows[:, :nth].reshape(n, nth, nth).transpose(0, 2, 1)),
                    allrows[:, nth].astype(np.int32).reshape(n, nth))
        lo, hi = block_partition(sim_begin, sim_end, self.world, self.rank)
        if hi > lo:
            Hs, its = self.local.implicit_H_batch(rng, lo, hi, theta0, atol=atol, cg_maxiter=cg_maxiter)
        else:
            Hs, its = np.zeros((0, nth, nth)), np.zeros((0, nth), dtype=np.int32)
        counts = [block_partition(sim_begin, sim_end, self.world, r) for r in range(self.world)]
        counts = [h - l for l, h in counts]
        rows = np.concatenate([Hs.reshape(hi - lo, nth * nth), its.astype(np.float64).reshape(hi - lo, nth)], axis=1)
        allrows = self._allgather_rows(rows, counts)
        n = allrows.shape[0]
        return (np.ascontiguousarray(allrows[:, : nth * nth].reshape(n, nth, nth)),
                allrows[:, nth * nth:].astype(np.int32))

    # -- resident MAPs.  Slots follow the element order of the LAST map: [data] + sims, the data element and the first
    #    block on rank 0, every rank's block in its local slots from 0 (rank 0: after the data element).
    def _slot_owner(self, nslots, had_data=True):
        """(rank, local slot) of every global slot 0..nslots-1 of a map of nslots elements: [data] + sims."""
        if nslots is None:
            raise ValueError("no sharded map has run on this problem yet (and no set_zhat): pass nslots, the element "
                             "count of the map whose MAPs are meant")
        d = 1 if had_data else 0
        nsims = nslots - d
        owners = [(0, 0)] if had_data else []
        for r in range(self.world):
            lo, hi = block_partition(0, nsims, self.world, r)
            owners += [(r, (s - lo) + (d if r == 0 else 0)) for s in range(lo, hi)]
        return owners

    def get_zhat(self, slot_begin, slot_end, nslots=None):
        """MAPs of global slots [slot_begin, slot_end) of the last (nsims+1)-element muse! map, gathered from the owning
        ranks (save_MAPs, src/muse.jl:139-143,219).  nslots = nsims + 1 of that map (default: what the last sharded
        map_and_score_batch with include_data used)."""
        had_data = self._last_had_data if nslots is None else True
        nslots = self._last_nslots if nslots is None else nslots
        owners = self._slot_owner(nslots, had_data)[slot_begin:slot_end]
        mine = [ls for (r, ls) in owners if r == self.rank]
        rows = np.zeros((len(mine), self.local.N))
        for k, ls in enumerate(mine):
            rows[k] = self.local.get_zhat(ls, ls + 1)[0]
        counts = [sum(1 for (r, _) in owners if r == q) for q in range(self.world)]
        allrows = self._allgather_rows(rows, counts)      # concatenated in rank order = slot order (blocks are contiguous)
        return allrows

    def set_zhat(self, slot_begin, zs, nslots=None):
        """Starting guesses for global slots slot_begin.. (z₀ of muse!, src/muse.jl:151): every rank keeps its own rows."""
        zs = np.atleast_2d(np.asarray(zs, dtype=np.float64))
        nslots = slot_begin + zs.shape[0] if nslots is None else nslots
        self._last_nslots = nslots
        self._last_had_data = True
        for k, (r, ls) in enumerate(self._slot_owner(nslots)[slot_begin:slot_begin + zs.shape[0]]):
            if r == self.rank:
                self.local.set_zhat(ls, zs[k:k + 1])
