"""Register budget of the hot kernel instantiations: the resident kernels of the headline config
must not spill vector registers to scratch.  Compiles each instantiation alone to assembly (hipcc cross-compiles
without a GPU; a few seconds each)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_hot_kernels_do_not_spill():
    spec = importlib.util.spec_from_file_location("regs", os.path.join(ROOT, "tools", "regs.py"))
    regs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(regs)
    assert regs.check() == []


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"), reason="llvm-readelf not available")
def test_built_libraries_keep_the_solver_state_out_of_scratch():
    """The code object of the BUILT libraries (what runs, not a single-instantiation build): no solver kernel calls a device
    function or keeps more than a few hundred bytes of scratch per lane -- the inliner of the library build once left
    Solver::run as a function of its own for FunnelModel<8> (1 KB of scratch per lane: the solver's state, its register-resident
    vectors included, behind `this`; 189 us per 512-sim step against 67 us for FunnelModel<4>); and the headline kernels of the
    product library spill nothing there either."""
    import glob
    import museinference_jl_amd as M
    spec = importlib.util.spec_from_file_location("regs", os.path.join(ROOT, "tools", "regs.py"))
    regs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(regs)
    main = M.build.build_extension() if hasattr(M, "build") else os.path.join(ROOT, "museinference.jl_amd", "libmuse_hip.so")
    libs = [main] + sorted(glob.glob(os.path.join(ROOT, "museinference.jl_amd", "libmuse_hip_model_cubic.so")))
    for lib in libs:
        assert regs.check_library(lib) == [], lib
    rows = {r[0]: r for r in regs.library_report(main)}
    for key in ("11FunnelModelILi1EEENS_13PlaceResidentILi512ELi10ELb1ELb0EEELb0EE", "10NoiseModelENS_13PlaceResidentILi512ELi10ELb1ELb0EEELb0EE"):
        assert rows[key][2] == 0 and rows[key][4] == 0, rows[key]     # (vgpr_spill_count, scratch bytes)
    # round 5: the loop kernel muse() runs by default -- with the MAP kept in registers from one iteration to the next -- too
    for key in ("loop:11FunnelModelILi1EEENS_13PlaceResidentILi512ELi10ELb1ELb0EEEENS_8LoopArgsE",
                "loop:10NoiseModelENS_13PlaceResidentILi512ELi10ELb1ELb0EEEENS_8LoopArgsE"):
        assert rows[key][2] == 0 and rows[key][4] == 0, rows[key]
    assert len(rows) > 90     # (round 5 dropped the loop kernels of the streaming placements, which nothing launched)
