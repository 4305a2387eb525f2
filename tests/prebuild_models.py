"""The model libraries the GPU tests and examples/user_model.py ask for by source text (ElementwiseModel.from_source: a library
per hash of the generated header, ~40 s of hipcc each): `models()` lists them so that __graft_entry__.build() can compile them
ahead of time next to the product library -- a GPU test run then loads them instead of compiling on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def models():
    import museinference_jl_amd as M
    import test_user_model as T
    text = open(os.path.join(ROOT, "tests", "test_user_model.py")).read()
    soft = text.split("src = '''")[1].split("'''")[0]
    bad = soft.replace('"softprior"', '"badpad"').replace("return r * r;", "return r * r + 1.0;")
    example = open(os.path.join(ROOT, "examples", "user_model.py")).read().split("SOURCE = r'''")[1].split("'''")[0]
    out = [M.ElementwiseModel.from_source("spectrum", T.SPECTRUM_SOURCE, constants={"P": T.spectrum(N)}) for N in T.SPECTRUM_SIZES]
    import numpy as np
    spec = open(os.path.join(ROOT, "examples", "spectrum.py")).read()
    spec_src = spec.split("SOURCE = r'''")[1].split("'''")[0]
    P = 30.0 / (1.0 + np.arange(10000) % 250) ** 1.7 + 0.02          # (examples/spectrum.py's table)
    out.append(M.ElementwiseModel.from_source("known_spectrum", spec_src, constants={"P": P}))
    out.append(M.ElementwiseModel.from_source("spectrum", T.SPECTRUM_SOURCE, runtime_constants=["P"]))   # (one library for every N and P)
    out.append(M.ElementwiseModel.from_source("noise_second", T.NOISE_SECOND_SOURCE))
    import test_symbolic_model as TS
    out.append(M.ElementwiseModel("pair_heavy_score", TS.HEAVY))   # (a correct header whose loop kernels exceed the scratch bound)
    out.append(TS.generated_nmv(M))                       # (... of the two-parameter family)
    out.append(TS.generated_cubic(M))                     # (a header generated from the model's terms: museinference_jl_amd.symbolic)
    terms = open(os.path.join(ROOT, "examples", "model_from_terms.py")).read().split("TERMS = dict(")[1].split(")\n")[0]
    out.append(M.ElementwiseModel.from_expressions("growing_response", **eval("dict(" + terms + ")")))      # (examples/model_from_terms.py)
    out += [M.ElementwiseModel.from_source("softprior", soft), M.ElementwiseModel.from_source("badpad", bad),
            M.ElementwiseModel.from_source("saturating", example)]
    return out


if __name__ == "__main__":
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(4) as pool:
        print(list(pool.map(lambda m: m.library(), models())))
