"""The parity helper itself (tests/test_gpu_parity.py: assert_same_path_or_close), on made-up solver records, without a GPU: both of
its branches -- equal iteration / evaluation counts => the tight tolerances; different counts => the bound the MAP tolerance implies,
|dz|_inf <= 2 atol / lambda_min -- must RUN (round 5's review found a call whose arguments were swapped in a way only the second
branch could notice)."""
import numpy as np
import pytest

from test_gpu_parity import assert_same_path_or_close

INFO = np.dtype([("iterations", "<i4"), ("f_calls", "<i4"), ("status", "<i4"), ("hist_words", "<i4"), ("f_min", "<f8"), ("gnorm", "<f8")])


def infos(iters, fcalls):
    a = np.zeros(len(iters), dtype=INFO)
    a["iterations"], a["f_calls"] = iters, fcalls
    return a


@pytest.mark.parametrize("model,theta,lam", [("funnel", [0.5], 1.0), ("smooth", [1.0, 2.0], float(np.exp(-2.0)))])
def test_off_path_branch_applies_the_bound(model, theta, lam):
    rng = np.random.default_rng(3)
    N, atol = 50, 1e-3
    zo = rng.normal(size=(3, N))
    go = rng.normal(size=(3, len(theta)))
    io = infos([5, 7, 9], [11, 15, 20])
    info = infos([5, 8, 9], [11, 17, 20])          # element 1 left the oracle's path
    z = zo.copy()
    z[1] += 1.5 * atol / lam                        # inside 2 atol / lambda_min, far outside the same-path 1e-9
    g = go.copy()
    g[1] += 1e-3
    same = assert_same_path_or_close(info, io, z, zo, g, go, atol, theta, model, "made up")
    assert same.tolist() == [True, False, True]
    z[1] += 1.0 * atol / lam                        # 2.5 atol / lambda_min: beyond the bound
    with pytest.raises(AssertionError):
        assert_same_path_or_close(info, io, z, zo, g, go, atol, theta, model, "made up")
    z[1] = zo[1]
    z[0] += 1e-6                                    # a same-path element off by more than z_atol
    with pytest.raises(AssertionError):
        assert_same_path_or_close(info, io, z, zo, g, go, atol, theta, model, "made up")


def test_swapped_arguments_are_noticed():
    """(atol, theta, model) in any other order fails loudly on the off-path branch instead of passing on the other one."""
    zo = np.zeros((2, 8))
    io, info = infos([3, 4], [7, 9]), infos([3, 5], [7, 9])
    with pytest.raises(TypeError):
        assert_same_path_or_close(info, io, zo, zo, None, None, "funnel", [0.1], 1e-2, "swapped")
