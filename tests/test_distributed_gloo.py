"""The N>1 path on CPU: two processes, gloo backend, the oracle as the local engine.  The sharded
result must equal the single-process result bit for bit (every rank reduces the gathered scores in the
reference's sim order)."""
import os
import pickle
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, pickle
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch.distributed as dist
import museinference_jl_amd as M
from oracle import oracle as O
from oracle_problem import OracleBatchedProblem
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=world)
x, _ = O.sample_x_z("funnel", 256, 5, M.DATA_SIM, [0.0, 0.0])
local = OracleBatchedProblem(x, "funnel", 2, prior=M.GaussianPrior(0.0, 3.0), nthreads=1)
assert M.ranks_share_node(dist) is True          # one kernel, one /dev/shm: what the shared-memory transport needs
prob = M.ShardedMuseProblem(local)
assert prob.engine_comm is False and prob.transport is None     # the oracle stand-in has no engine communicator: torch collectives
try:
    M.ShardedMuseProblem(local).get_zhat(0, 1)
    raise SystemExit("get_zhat before any sharded map must raise")
except ValueError as e:
    assert "nslots" in str(e)
res = M.muse(prob, [1.0, 0.5], rng=3, nsims=13, maxsteps=4, get_covariance=True)
g, info = prob.map_and_score_batch(3, 2, 9, [0.1, 0.2], include_data=True)
Hi, its = prob.implicit_H_batch(3, 0, 5, [0.1, 0.2])
# get_H! with few sims and several theta: the (sim, column) units are shared, not the sims (src/muse.jl:327-333)
x4, _ = O.sample_x_z("funnel", 200, 6, M.DATA_SIM, [0.0] * 4)
local4 = OracleBatchedProblem(x4, "funnel", 4, prior=M.GaussianPrior(0.0, 3.0), nthreads=1)
prob4 = M.ShardedMuseProblem(local4)
H4, _ = prob4.fd_jacobian_batch(9, 0, 3, [0.3, 0.1, -0.2, 0.5], [0.05] * 4, atol=1e-3, fid_mode=0)
Hi4, its4 = prob4.implicit_H_batch(9, 0, 3, [0.3, 0.1, -0.2, 0.5])
# save_MAPs and a starting guess z0 under sharding: slots are gathered from / scattered to their owners
r2 = M.muse(prob, [1.0, 0.5], rng=3, nsims=5, maxsteps=2, save_MAPs=True, z0=np.full(256, 0.05))
with open({out!r} + str(rank), "wb") as f:
    pickle.dump(dict(theta=res.theta, J=res.J, H=res.H, Sigma=res.Sigma, gs=np.array(res.gs), g=g,
                     iters=info["iterations"], nlocal=len(local._zhat), Hi=Hi, its=its, H4=H4, Hi4=Hi4, its4=its4,
                     fd_maps=local4.fd_maps_done, zdat=r2.history[-1]["ẑ_dat"], zsims=np.array(r2.history[-1]["ẑ_sims"]),
                     theta2=r2.theta), f)
dist.destroy_process_group()
"""


def test_two_rank_gloo_matches_single_process(tmp_path):
    import museinference_jl_amd as M
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_problem import OracleBatchedProblem
    out = str(tmp_path / "res")
    port = 29500 + os.getpid() % 2000
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, port=port, out=out))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = [pickle.load(open(out + str(r), "rb")) for r in range(2)]
    x, _ = O.sample_x_z("funnel", 256, 5, M.DATA_SIM, [0.0, 0.0])
    single = OracleBatchedProblem(x, "funnel", 2, prior=M.GaussianPrior(0.0, 3.0), nthreads=1)
    ref = M.muse(single, [1.0, 0.5], rng=3, nsims=13, maxsteps=4, get_covariance=True)
    gref, iref = single.map_and_score_batch(3, 2, 9, [0.1, 0.2], include_data=True)
    Hiref, itsref = single.implicit_H_batch(3, 0, 5, [0.1, 0.2])
    for r in range(2):
        for k, want in (("theta", ref.theta), ("J", ref.J), ("H", ref.H), ("Sigma", ref.Sigma),
                        ("gs", np.array(ref.gs)), ("g", gref), ("iters", iref["iterations"]),
                        ("Hi", Hiref), ("its", itsref)):
            assert np.array_equal(got[r][k], want), (r, k)
    # each rank solved only its own block (data element on rank 0)
    assert got[0]["nlocal"] + got[1]["nlocal"] >= 8 and got[1]["nlocal"] < 14
    # get_H! at nsims = 3, ntheta = 4: 12 (sim, column) units, 6 per rank = 12 perturbed MAPs each; same H bit for bit
    x4, _ = O.sample_x_z("funnel", 200, 6, M.DATA_SIM, [0.0] * 4)
    s4 = OracleBatchedProblem(x4, "funnel", 4, prior=M.GaussianPrior(0.0, 3.0), nthreads=1)
    H4, _ = s4.fd_jacobian_batch(9, 0, 3, [0.3, 0.1, -0.2, 0.5], [0.05] * 4, atol=1e-3, fid_mode=0)
    Hi4, its4 = s4.implicit_H_batch(9, 0, 3, [0.3, 0.1, -0.2, 0.5])
    single2 = OracleBatchedProblem(x, "funnel", 2, prior=M.GaussianPrior(0.0, 3.0), nthreads=1)
    r2 = M.muse(single2, [1.0, 0.5], rng=3, nsims=5, maxsteps=2, save_MAPs=True, z0=np.full(256, 0.05))
    for r in range(2):
        assert got[r]["fd_maps"] == 12
        assert np.array_equal(got[r]["H4"], H4) and np.array_equal(got[r]["Hi4"], Hi4) and np.array_equal(got[r]["its4"], its4)
        assert np.array_equal(got[r]["theta2"], r2.theta)
        assert np.array_equal(got[r]["zdat"], r2.history[-1]["ẑ_dat"])
        assert np.array_equal(got[r]["zsims"], np.array(r2.history[-1]["ẑ_sims"]))


FALLBACK_WORKER = r"""
import os, sys, pickle
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch.distributed as dist
import museinference_jl_amd as M
from oracle import oracle as O
from oracle_problem import OracleBatchedProblem
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=world)


class FlakyEngineProblem(OracleBatchedProblem):
    # a local problem that HAS an engine communicator, which comes up on rank 0 and fails on rank 1
    log = []

    @staticmethod
    def comm_unique_id(transport="rccl", block_doubles=0):
        return bytes(128)

    def comm_init(self, nranks, rank_, uid):
        self.log.append(("init", rank_))
        if rank_ == 1:
            raise RuntimeError("segment not visible on this rank")
        self._nranks = nranks

    def comm_destroy(self):
        self.log.append(("destroy",))
        self._nranks = None

    def allgather_scores(self, send):
        raise AssertionError("the engine communicator must not be used after the ranks agreed it failed")


x, _ = O.sample_x_z("funnel", 128, 5, M.DATA_SIM, [0.0])
local = FlakyEngineProblem(x, "funnel", 1, prior=M.GaussianPrior(0.0, 3.0), nthreads=1)
prob = M.ShardedMuseProblem(local)          # candidates: shm (the ranks share this node); it fails on rank 1 -> all fall back
assert prob.engine_comm is False and prob.transport is None, (prob.engine_comm, prob.transport)
assert ("destroy",) in local.log if rank == 0 else ("destroy",) not in local.log     # rank 0 gave its communicator back
g, info = prob.map_and_score_batch(3, 0, 7, [0.4], include_data=True)              # torch.distributed collectives
try:
    M.ShardedMuseProblem(FlakyEngineProblem(x, "funnel", 1, nthreads=1), engine_comm=True, transport="shm")
    raise SystemExit("an explicitly requested transport that fails must raise on every rank")
except RuntimeError as e:
    assert "could not be initialised" in str(e)
with open({out!r} + str(rank), "wb") as f:
    pickle.dump(dict(g=g), f)
dist.destroy_process_group()
"""


def test_engine_transport_failure_is_a_collective_decision(tmp_path):
    """a transport that fails to come up on ONE rank (a /dev/shm the rank cannot see, a communicator that times
    out) must not leave the other ranks waiting or the run dead: every rank reports, all of them give the transport up
    together and move on (here: to torch.distributed collectives), and the sharded map still equals the single-process one."""
    import museinference_jl_amd as M
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_problem import OracleBatchedProblem
    out = str(tmp_path / "res")
    port = 31500 + os.getpid() % 2000
    script = tmp_path / "worker.py"
    script.write_text(FALLBACK_WORKER.format(root=ROOT, port=port, out=out))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                                                                     OMP_NUM_THREADS="1")) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    x, _ = O.sample_x_z("funnel", 128, 5, M.DATA_SIM, [0.0])
    single = OracleBatchedProblem(x, "funnel", 1, prior=M.GaussianPrior(0.0, 3.0), nthreads=1)
    gref, _ = single.map_and_score_batch(3, 0, 7, [0.4], include_data=True)
    for r in range(2):
        assert np.array_equal(pickle.load(open(out + str(r), "rb"))["g"], gref)
