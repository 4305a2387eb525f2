#!/bin/bash
# Interleaved A/B of the in-tree library against museinference.jl_amd/libmuse_hip_old.so (a variant build) on ONE box: the pipelined
# cold step of the given workloads (bench.py) and the native muse! loops at configs[1] and at its 8-GPU share (tools/runloop_bench.py).
# usage (inside gpurun): bash tools/ab_loops.sh "<workloads>" [reps]
WL=${1:-"funnel_1e4"}; REPS=${2:-2}
for rep in $(seq $REPS); do for lib in "" museinference.jl_amd/libmuse_hip_old.so; do
  if [ -n "$lib" ]; then export MUSE_HIP_LIB=$PWD/$lib; else unset MUSE_HIP_LIB; fi
  for w in $WL; do python bench.py --steps 100 --warmup 20 --min-seconds 0.3 --no-cpu-baseline --no-extra --workload $w 2>/dev/null | python tools/benchline.py "lib=${lib:-new} $w"; done
  python tools/runloop_bench.py 10000 512 1 dev 2>/dev/null | awk -v l="lib=${lib:-new}" "{print l, \$0}"
  python tools/runloop_bench.py 10000 64 1 dev 2>/dev/null | awk -v l="lib=${lib:-new}" "{print l, \$0}"
done; done
