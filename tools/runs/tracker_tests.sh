#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_tracker_gpu.py tests/test_golden_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r02_tracker_tests.log 2>&1; tail -5 gpurun_out/r02_tracker_tests.log
