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


def test_library_exports_exactly_the_header(M):
    """`nm -D` of the product library = the entry points include/muse_hip.h declares, nothing else: the accessors between the
    library's own translation units, the kernels' host stubs and the C++ runtime's weak symbols are local (a linker version
    script written from the header, build.py)."""
    import subprocess
    path = M.build_extension()
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    exported = sorted({line.split()[-1].split("@")[0] for line in out.splitlines() if line.strip() and " A " not in line})
    assert exported == declared_symbols(), (sorted(set(exported) - set(declared_symbols())), sorted(set(declared_symbols()) - set(exported)))


def test_no_getenv_outside_context_creation():
    """The environment switches are read once per context (csrc/switches.hpp, from muse_ctx_create): no other getenv in the library."""
    csrc = os.path.join(ROOT, "museinference.jl_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        text = open(os.path.join(csrc, f)).read()
        text = re.sub(r"//[^\n]*", "", text)
        if f == "switches.hpp":
            assert "getenv" in text
        else:
            assert "getenv" not in text, f
    eng = open(os.path.join(csrc, "muse_engine.cpp")).read()
    assert eng.count("Switches::from_environment()") == 1


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


# ---- the two boundary documents against the Julia shim ("boundary drift") ---------------------------------------
def _integration_rows():
    rows = []
    for line in open(os.path.join(ROOT, "INTEGRATION.md")):
        if line.startswith("| `muse_"):
            cells = [c.strip() for c in line.strip().strip("|").split("|")]
            rows.append((cells[0], cells[3]))
    return rows


def _expand(cell):
    """C symbols a table cell names: `muse_a/b`, `muse_x[_y]` and `muse_comm_init/destroy/...` spelled out."""
    out = []
    for tok in re.findall(r"`([^`]+)`", cell):
        if not tok.startswith("muse_"):
            continue
        m = re.match(r"^(muse_[a-z_A-Z0-9]*?)\[(_[a-z_A-Z0-9]+)\]$", tok)
        if m:
            out += [m.group(1), m.group(1) + m.group(2)]
            continue
        parts = tok.split("/")
        out.append(parts[0])
        stem = parts[0][: parts[0].rfind("_") + 1]
        for p in parts[1:]:
            out.append(p if p.startswith("muse_") else stem + p)
    return out


def test_julia_shim_binds_what_integration_md_says_it_binds():
    """Every row of INTEGRATION.md's table whose Julia column names a binding (anything but "Python binding only" / "--") must
    have at least one of its C entry points `ccall`ed by julia/HipMuseInference.jl, the entry points the round-4 review found
    missing each by name; every Julia function the column names must be defined there; and the shim `ccall`s nothing that
    include/muse_hip.h does not declare."""
    shim = open(os.path.join(ROOT, "julia", "HipMuseInference.jl")).read()
    called = set(re.findall(r"ccall\(\(:(muse_[a-z_A-Z0-9]+), libmuse_hip\)", shim))
    declared = set(declared_symbols())
    assert called <= declared, sorted(called - declared)
    rows = _integration_rows()
    assert len(rows) >= 20
    for csyms, jl in rows:
        names = [n for n in _expand(csyms) if n in declared]
        assert names, csyms
        whole_row_python_only = jl in ("Python binding only", "—")
        if whole_row_python_only:
            continue
        assert any(n in called for n in names), (csyms, jl)
        for fn in re.findall(r"`([A-Za-z_][A-Za-z_0-9!]*)\(", jl):     # `name(...)` in the Julia column: a function of the shim
            if fn in ("HipMuseProblem",):
                continue
            assert re.search(r"(^|\n)(function )?%s\(" % re.escape(fn), shim), (fn, csyms)
    for must in ("muse_run_device", "muse_run_sharded", "muse_run", "muse_set_constants", "muse_set_normals_cache", "muse_implicit_H_batch",
                 "muse_implicit_H_columns", "muse_model_eval", "muse_model_has_second", "muse_fd_values_columns", "muse_fd_jacobian_columns"):
        assert must in called, must
