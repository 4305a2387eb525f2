"""museinference.jl_amd -- MI355X-native engine for the MUSE inner loop.

The hot path of marius311/MuseInference.jl (per-simulation sample -> latent MAP by L-BFGS -> score,
batched over all simulations of muse!/get_J!/get_H!) as hand-written HIP kernels behind a C ABI
(include/muse_hip.h, libmuse_hip.so), plus the host-side mirror of the reference's problem interface
and drivers.  See DESIGN.md and INTEGRATION.md at the repository root.
"""
from . import _capi
from ._capi import MuseError, STATUS_NAMES, Z0_TRUE, Z0_WARM, Z0_ZERO, load_library
from .build import build_extension
from .covariance import LinearShrinkage, SimpleCovariance
from .fdm import FiniteDifferenceMethod, central_fdm
from .models import ElementwiseModel, check_model_consistency
from .distributed import ShardedMuseProblem, block_partition, ranks_share_node
from .muse import (MuseResult, Normal, finalize_result_, get_H_, get_J_, load_result, muse, muse_, save_result)
from .priors import CallablePrior, FlatPrior, GaussianPrior
from .simple import TorchMuseProblem
from .problem import (DATA_SIM, MASTER_SIM, AbstractMuseProblem, HipMuseProblem, PositiveThetaProblem, SimRng,
                      Transformedθ, UnTransformedθ, check_optim_soln, check_self_consistency, split_rng)

__all__ = [
    "AbstractMuseProblem", "HipMuseProblem", "ShardedMuseProblem", "MuseResult", "Normal", "SimRng",
    "muse", "muse_", "get_J_", "get_H_", "finalize_result_", "split_rng", "block_partition", "ranks_share_node", "central_fdm", "FiniteDifferenceMethod",
    "ElementwiseModel", "check_model_consistency", "SimpleCovariance", "LinearShrinkage", "TorchMuseProblem", "GaussianPrior", "FlatPrior", "CallablePrior", "build_extension", "load_library", "MuseError",
    "save_result", "load_result", "check_optim_soln", "check_self_consistency", "PositiveThetaProblem", "Transformedθ", "UnTransformedθ",
    "Z0_ZERO", "Z0_TRUE", "Z0_WARM", "MASTER_SIM", "DATA_SIM", "STATUS_NAMES",
]
