"""Registers / spills / scratch of the solver kernel variants, from the device assembly.

  python tools/regs.py file.s           report every map_score_kernel variant of an assembly file
                                        (hipcc -S --cuda-device-only ... muse_kernels.hip -o file.s: ~2.5 min for all)
  python tools/regs.py --library lib.so report the solver kernels of a BUILT library from its own code object, and fail if one calls a
                                        device function or keeps the solver's state in scratch (what the library build can do and a
                                        single-instantiation build does not show)
  python tools/regs.py --check          compile ONLY the hot instantiations (-DMUSE_INSPECT=..., a few seconds each)
                                        and fail if one of them spills vector registers to scratch
The hot instantiations: the resident kernels of BASELINE.json configs[1] (funnel, 1 theta) and of the noise model
at N <= 10^4 must have vgpr_spill_count == 0 (a spilled value comes back from scratch in the line search and in
every reduction); the other variants are reported with their budget."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "museinference.jl_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-S", "--cuda-device-only"]
# (instantiation, largest tolerated vgpr_spill_count)
HOT = [
    ("FunnelModel<1>, PlaceResident<512, 10, true>", 0),   # configs[1], the headline
    ("NoiseModel, PlaceResident<512, 10, true>", 0),
    # configs[3] (4 theta blocks) and its per-GPU share under an element split of 4, and the 8-block variant: round 2 had
    # 21 / 13 / 33 spilled VGPRs and 172 / 114 / 353 spilled SGPRs here -- lane masks and LDS addresses derived from the packed
    # block indices, hoisted to the kernel's entry (solver.hpp, blk()).  What is left is a handful of values stored and
    # re-loaded ONCE per problem (nothing inside an element loop): tolerated up to the counts below.
    ("FunnelModel<4>, PlaceResident<512, 10, true>", 2),
    ("FunnelModel<4>, PlaceResident<512, 3, false, true>", 16),
    ("FunnelModel<8>, PlaceResident<512, 10, true>", 12),
    ("FunnelModel<1>, PlaceResident<512, 3, false, true>", 6),
    # the background generator's sums across a pass: once per pass, not per trip; round 5: the speculating trial (solver.hpp, eval SPEC)
    # stages z + c s beside them -- 8 -> 15 spilled registers, and noise_1e6 1.303 -> 1.048 ms per step on one box (a pass over HBM less)
    ("NoiseModel, PlaceStreaming<256, true>", 16),
    ("SmoothModel<8>, PlaceStreaming<256, true, 2, true, true>", 0),   # configs[4]: clusters with the direction in LDS
]


# the device-resident muse! loop (muse_loop_kernel): the same budget as the map kernel of the placement
# (what is tolerated: values stored at the kernel's entry and re-loaded once per PROBLEM, between two problems -- none inside
# an element loop, a reduction or the line search)
# Round 5: the roles as two loops and ONE copy of the solve in the worker's loop (kernels.hpp) -- the LDS-resident loop kernels of one
# and two components are out of scratch altogether (12 -> 0 spilled VGPRs, with the MAP kept in registers across iterations), four
# components 37 -> 22, the all-register placement (N <= 4096) 100 -> 33-44.
HOT_LOOP = [
    ("FunnelModel<1>, PlaceResident<512, 10, true>", 0),
    ("NoiseModel, PlaceResident<512, 10, true>", 0),
    # (the multi-component counts move by a few registers with any change of the surrounding source -- 0-3, 22-25, 33-36, 36-40 over
    #  this round's builds: the limits leave that much room and no more)
    # (with the stepper's own copy of the solve -- a stepper that owns elements, later in round 5 -- the one-component kernels stay at 0
    #  and the others went up by 10-17: 4, 35, 55, 65; tools/loop_vs_host.py, the loop kernel against the host loop per iteration of a
    #  30-iteration call, 512 sims: N = 10^4 x 4: 69.4 / 66.7 us wall, 60.1 / 59.7 steady; N = 4096 x 4: 52.6 / 61.3 wall)
    ("FunnelModel<2>, PlaceResident<512, 10, true>", 8),
    # (round 6, the deferred store of the MAP a worker carries into the next iteration: 36 -> 49 at four components, and the iteration
    #  FASTER -- 62.1 against 64.3 us per iteration of a 30-iteration call, three alternating pairs; the spills sit between problems)
    ("FunnelModel<4>, PlaceResident<512, 10, true>", 52),
    ("FunnelModel<1>, PlaceResident<512, 4, false>", 60),
    ("FunnelModel<4>, PlaceResident<512, 4, false>", 70),
]


def report(path):
    rows = []
    txt = open(path).read()
    for b in txt.split('  - .agpr_count:')[1:]:
        name = re.search(r'\.name:\s+(\S+)', b).group(1)
        if 'map_score' not in name and 'muse_loop' not in name:
            continue
        g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, b).group(1))
        short = name.replace('_ZN4muse16map_score_kernelINS_', '').replace('_ZN4muse16muse_loop_kernelINS_', 'loop:').replace('EvNS_9BatchArgsE', '')
        rows.append((short, g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'), g('private_segment_fixed_size')))
    return rows


def compile_one(inst, out, loop=False):
    hipcc = "/opt/rocm/bin/hipcc"
    defs = ["-DMUSE_INSPECT=" + inst] + (["-DMUSE_INSPECT_LOOP=" + inst] if loop else [])
    subprocess.check_call([hipcc] + FLAGS + defs + [os.path.join(CSRC, "muse_kernels.hip"), "-o", out],
                          cwd=CSRC, stderr=subprocess.DEVNULL)


def check():
    bad = []
    with tempfile.TemporaryDirectory() as d:
        for inst, limit in HOT:
            out = os.path.join(d, "one.s")
            compile_one(inst, out)
            (short, vgpr, vspill, sspill, scratch), = report(out)
            ok = vspill <= limit
            print(f"{inst:55s} vgpr {vgpr:3d} vspill {vspill:3d} (limit {limit:2d}) sspill {sspill:3d} scratch {scratch:3d} {'ok' if ok else 'FAIL'}")
            if not ok:
                bad.append(inst)
        for inst, limit in HOT_LOOP:
            out = os.path.join(d, "loop.s")
            compile_one(inst, out, loop=True)
            rows = [r for r in report(out) if r[0].startswith("loop:")]
            (short, vgpr, vspill, sspill, scratch), = rows
            ok = vspill <= limit
            print(f"{'loop kernel: ' + inst:55s} vgpr {vgpr:3d} vspill {vspill:3d} (limit {limit:2d}) sspill {sspill:3d} scratch {scratch:3d} {'ok' if ok else 'FAIL'}")
            if not ok:
                bad.append("loop: " + inst)
    return bad


def library_report(path):
    """Every solver kernel of a BUILT library (libmuse_hip.so, a model's library), from the metadata of its gfx950 code object: the
    single-instantiation builds of --check do not show what the inliner does with a hundred instantiations in one translation
    unit (round 4: Solver::run left as a function of its own for some of them -- the solver's state, its register-resident vectors
    included, in memory behind `this`: 1 KB of scratch per lane and a 3x slower FunnelModel<8>)."""
    import struct
    data = open(path, "rb").read()
    # one offload bundle per device translation unit (round 5: the kernels are compiled as several units, side by side)
    blobs, start = [], 0
    while True:
        i = data.find(b"__CLANG_OFFLOAD_BUNDLE__", start)
        if i < 0:
            break
        n, = struct.unpack_from("<Q", data, i + 24)
        pos, end = i + 32, i + 32
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", data, pos)
            pos += 24
            tid = data[pos:pos + idlen].decode()
            pos += idlen
            if "gfx950" in tid:
                blobs.append(data[i + off:i + off + size])
            end = max(end, i + off + size)
        start = max(end, i + 24)
    if not blobs:
        raise RuntimeError(path + ": no gfx950 code object found")
    out = ""
    for blob in blobs:
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            out += subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True, check=True).stdout
    rows = []
    for b in out.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        if "map_score" not in name and "muse_loop" not in name:
            continue
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, b).group(1))
        short = name.replace("_ZN4muse16map_score_kernelINS_", "").replace("_ZN4muse16muse_loop_kernelINS_", "loop:").replace("EvNS_9BatchArgsE", "")
        rows.append((short, g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"),
                     bool(re.search(r"\.uses_dynamic_stack:\s+true", b))))
    return rows


LIBRARY_SCRATCH_LIMIT = 256   # bytes per lane; the solver's state behind a pointer is > 1000 (round 4: 512; what is left above 128 are the
                              # two-workgroup register placements of an element split, 156-216 bytes, and nothing the default routing launches)


def check_library(path):
    """Kernels of a built library that call a device function (dynamic stack) or keep more than LIBRARY_SCRATCH_LIMIT bytes of
    scratch per lane."""
    return [r for r in library_report(path) if r[5] or r[4] > LIBRARY_SCRATCH_LIMIT]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        sys.exit(1 if check() else 0)
    if len(sys.argv) > 2 and sys.argv[1] == "--library":
        for short, vgpr, vspill, sspill, scratch, dyn in sorted(library_report(sys.argv[2]), key=lambda r: -r[4]):
            print(f"{short:78s} vgpr {vgpr:3d} vspill {vspill:3d} sspill {sspill:3d} scratch {scratch:4d}{' CALLS' if dyn else ''}")
        bad = check_library(sys.argv[2])
        print(f"{len(bad)} kernels beyond {LIBRARY_SCRATCH_LIMIT} bytes of scratch or with calls")
        sys.exit(1 if bad else 0)
    for short, vgpr, vspill, sspill, scratch in report(sys.argv[1]):
        print(f"{short:70s} vgpr {vgpr:3d} vspill {vspill:3d} sspill {sspill:3d} scratch {scratch}")
