#!/bin/bash
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r2_t17.log 2>&1; tail -3 gpurun_out/r2_t17.log
A="--steps 10 --repeats 1 --no-cpu-baseline --no-extras"
bash tools/profile_timeline.sh r2_tl6 $A
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-400
