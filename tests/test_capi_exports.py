"""The C-ABI library loads on a GPU-less host and exports every symbol include/muse_hip.h declares
(no compute calls); the product path fails loudly without a device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "muse_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(muse_[a-z_A-Z0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(M):
    path = M.build_extension()
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/muse_hip.h but not exported"
    # and the ctypes binding covers exactly the header
    assert sorted(M._capi.SIGNATURES) == names


def test_no_cpu_fallback(M):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(M.MuseError) as e:
        M.HipMuseProblem(None, N=16)
    assert "no HIP device" in str(e.value)


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "museinference.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                # comments may cite the oracle; code must not import, link, load or include it
                for pat in ("import oracle", "from oracle", "libmuse_oracle", "muse_oracle.h", '#include "../../oracle',
                            "oracle.oracle", "mo_"):
                    if pat == "mo_":
                        import re
                        assert not re.search(r"\bmo_[a-z_]+\s*\(", src.replace("mo_implicit_H,", "")), f
                    else:
                        assert pat not in src, (f, pat)
