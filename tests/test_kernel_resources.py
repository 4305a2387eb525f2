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
