#!/bin/bash
# Development aid: registers / spills of ONE instantiation of the solver kernel and of the loop kernel (tools/regs.py reads the assembly).
#   tools/inspect_one.sh 'FunnelModel<1>, PlaceResident<512, 10, true>' [extra hipcc flags]
# (a placement the loop kernel is not built for: set NOLOOP=1)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
inst="$1"; shift
out=/tmp/inspect_$(echo "$inst" | tr -c 'A-Za-z0-9' '_').s
loopdef=()
[ -z "$NOLOOP" ] && loopdef=("-DMUSE_INSPECT_LOOP=$inst")
(cd "$ROOT/museinference.jl_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -S --cuda-device-only \
    "-DMUSE_INSPECT=$inst" "${loopdef[@]}" "$@" muse_kernels.hip -o "$out" 2>&1 | grep -v "warning: argument unused" || true)
python "$ROOT/tools/regs.py" "$out"
