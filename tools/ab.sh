#!/bin/bash
# A/B on ONE box: the in-tree library against tools/libmuse_prev.bin (a build of another revision), interleaved
for r in 1 2 3; do
for lib in cur prev; do
  if [ $lib = prev ]; then export MUSE_HIP_LIB=$PWD/tools/libmuse_prev.bin; else unset MUSE_HIP_LIB; fi
  for w in ${@:-funnel_1e4}; do
    python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra --workload $w 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '$w', round(d['value']), round(1e3*d['roofline']['kernel_ms_mean'],2), 'us')"
  done
done
done
