mkdir -p gpurun_out/r05u
python -m pytest tests -m gpu -x -q > gpurun_out/r05u/pytest_gpu.log 2>&1
python bench.py > gpurun_out/r05u/bench_default.json 2> gpurun_out/r05u/bench_default.err
python tools/share_bench.py > gpurun_out/r05u/share_bench.log 2>&1
