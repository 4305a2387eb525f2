"""csrc/step.hpp on CPU (a g++ build of tests/native/step_driver.cpp): the diagonal closed form of the muse! step that the
library runs -- on the host in muse_run / muse_run_sharded, on one lane of the GPU in muse_run_device -- against its
definition with dense Gauss-Jordan inverses (src/muse.jl:208 inverts general matrices): the same bits, signed zeros,
infinities, NaNs and error codes included; the 64-leaf summation tree of the score moments; and block_of_big, the arithmetic
block index of the big tier (ntheta > MUSE_MAX_THETA), against its integer definition."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_step_closed_form_equals_dense_definition_bitwise(tmp_path):
    exe = str(tmp_path / "step_driver")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                           "-o", exe, os.path.join(HERE, "native", "step_driver.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "step driver ok" in r.stdout
    assert "block_of_big:" in r.stdout      # (the big tier's block index against floor(i B / N): tens of millions of elements)
