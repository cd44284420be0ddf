#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_golden_gpu.py tests/test_tracker_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r2_t29.log 2>&1; tail -2 gpurun_out/r2_t29.log
python tools/sia_stamps.py euroc > gpurun_out/r2_stamps29.log 2>&1; grep -v amdgpu.ids gpurun_out/r2_stamps29.log | grep -E "solve|kernel_total|pose\+key"
