#!/bin/bash
timeout -k 10 1100 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=12 > gpurun_out/r2_t24.log 2>&1; tail -22 gpurun_out/r2_t24.log
