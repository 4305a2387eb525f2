"""examples/: the reference's documentation example and a user-supplied model, run as a user would run them."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("script", ["quickstart.py", "user_model.py", "spectrum.py", "mean_and_variance.py", "model_from_terms.py", "closures.py"])
def test_example_runs(gpu, script):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "theta" in p.stdout
    print(p.stdout)
