#!/bin/bash
# scratch GPU-box script of round 2
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_tracker_gpu.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r2_t4.log 2>&1; tail -8 gpurun_out/r2_t4.log
python tools/sia_stamps.py euroc > gpurun_out/r2_stamps4.log 2>&1; python tools/sia_stamps.py euroc exact >> gpurun_out/r2_stamps4.log 2>&1; grep -v amdgpu.ids gpurun_out/r2_stamps4.log
bash tools/profile_bench.sh r2_p2 --seqs 256 --steps 20 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
bash tools/profile_bench.sh r2_p2x --seqs 256 --steps 20 --repeats 1 --no-cpu-baseline --no-extras --exact > /dev/null 2>&1
grep "svo::" gpurun_out/r2_p2/kernel_stats.csv | head -9
grep "svo::" gpurun_out/r2_p2x/kernel_stats.csv | head -5
cut -c1-400 gpurun_out/r2_p2/bench.json; echo; cut -c1-400 gpurun_out/r2_p2x/bench.json; echo
tail -3 gpurun_out/r2_p2/bench.err
timeout -k 10 500 python bench.py > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err; tail -3 gpurun_out/r2_bench1.err; cat gpurun_out/r2_bench1.json
