#!/bin/bash
# N>1 bench path exercised on ONE GPU (development aid): plain, engine-RCCL gather, torch.distributed gather
python -m pytest tests/test_gpu_drivers.py -m gpu -x -q -k "gathered or rccl" 2>&1 | tail -3
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open('/tmp/b.json').read())
print(sys.argv[1], round(d["value"]), round(1e3 * d["ms_per_step"], 1), "us/step; kernel", round(1e3 * d["roofline"]["kernel_ms_mean"], 1), "us; host", d["host_us_per_step"])
PY
}
B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline"
$B 2>&1 | grep "^{" | tail -1 > /tmp/b.json; show plain
MUSE_BENCH_FORCE_DIST=1 $B 2>&1 | grep "^{" | tail -1 > /tmp/b.json; show rccl
MUSE_BENCH_FORCE_DIST=1 MUSE_COMM_ONE_STREAM=1 $B 2>&1 | grep "^{" | tail -1 > /tmp/b.json; show rccl-one-stream
MUSE_BENCH_FORCE_DIST=1 MUSE_BENCH_COLLECTIVE=torch $B 2>&1 | grep "^{" | tail -1 > /tmp/b.json; show torch
MUSE_BENCH_FORCE_DIST=1 MUSE_COMM_DIRECT_HOST=1 $B 2>&1 | grep "^{" | tail -1 > /tmp/b.json; show rccl-directhost
MUSE_BENCH_FORCE_DIST=1 MUSE_COMM_DIRECT_HOST=1 MUSE_COMM_ONE_STREAM=1 $B 2>&1 | grep "^{" | tail -1 > /tmp/b.json; show rccl-directhost-onestream
