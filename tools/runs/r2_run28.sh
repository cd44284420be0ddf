#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_golden_gpu.py tests/test_tracker_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r2_t28.log 2>&1; tail -2 gpurun_out/r2_t28.log
python tools/sia_stamps.py euroc > gpurun_out/r2_stamps28.log 2>&1; grep -v amdgpu.ids gpurun_out/r2_stamps28.log | head -12
B="--no-cpu-baseline --no-extras --repeats 1"
echo "== svd micro"; timeout -k 10 300 python bench.py $B 2>/dev/null | cut -c1-120
echo "== again"; timeout -k 10 300 python bench.py $B 2>/dev/null | cut -c1-120
