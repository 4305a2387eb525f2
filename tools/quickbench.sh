#!/bin/bash
# tests + stamps + per-workload kernel times (development aid)
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/stamps.py 2>&1 | grep -v amdgpu.ids | head -9
for pl in -1 0; do python bench.py --steps 200 --warmup 20 --no-cpu-baseline --placement $pl 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('funnel_1e4 placement $pl', d['value'], d['roofline']['kernel_ms_mean'], d['roofline']['frac'])"; done
for w in funnel4_1e4 noise_1e6 smooth_1e5 funnel_512; do python bench.py --steps 5 --warmup 1 --workload $w --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:12], d['value'], d['roofline']['kernel_ms_mean'], d['roofline']['frac'])"; done
