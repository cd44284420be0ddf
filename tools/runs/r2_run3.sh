#!/bin/bash
# scratch GPU-box script of round 2 (tests + stamps + rocprof of both solver modes)
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_tracker_gpu.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r2_t3.log 2>&1; tail -12 gpurun_out/r2_t3.log
python tools/sia_stamps.py euroc > gpurun_out/r2_stamps.log 2>&1; python tools/sia_stamps.py euroc exact >> gpurun_out/r2_stamps.log 2>&1; python tools/sia_stamps.py econ >> gpurun_out/r2_stamps.log 2>&1; cat gpurun_out/r2_stamps.log
bash tools/profile_bench.sh r2_p1 --seqs 256 --steps 20 --no-cpu-baseline > /dev/null 2>&1
bash tools/profile_bench.sh r2_p1x --seqs 256 --steps 20 --no-cpu-baseline --exact > /dev/null 2>&1
grep -v "at::\|elementwise\|Memcpy\|rocrand\|vectorized" gpurun_out/r2_p1/kernel_stats.csv | head -14
grep -v "at::\|elementwise\|Memcpy\|rocrand\|vectorized" gpurun_out/r2_p1x/kernel_stats.csv | head -8
cut -c1-200 gpurun_out/r2_p1/bench.json; cut -c1-200 gpurun_out/r2_p1x/bench.json
