#!/bin/bash
# Interleaved A/B of the in-tree library against museinference.jl_amd/libmuse_hip_old.so (a variant build) on ONE box.
# usage (inside gpurun): bash tools/ab_libs.sh "<workloads>" [reps]
WL=${1:-"funnel_1e4 funnel4_1e4"}; REPS=${2:-2}
for rep in $(seq $REPS); do for lib in "" museinference.jl_amd/libmuse_hip_old.so; do
  if [ -n "$lib" ]; then export MUSE_HIP_LIB=$PWD/$lib; else unset MUSE_HIP_LIB; fi
  for w in $WL; do python bench.py --steps 50 --warmup 10 --min-seconds 0 --no-cpu-baseline --no-extra --workload $w 2>/dev/null | python tools/benchline.py "lib=${lib:-new} $w"; done
done; done
