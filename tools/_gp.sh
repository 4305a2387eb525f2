mkdir -p gpurun_out/r05x
python tools/loop_layouts.py > gpurun_out/r05x/layouts_1.log 2>&1
python tools/loop_layouts.py 10000 512 2 > gpurun_out/r05x/layouts_2.log 2>&1
