#!/bin/bash
# scratch GPU-box script of round 2
timeout -k 10 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r2_t7.log 2>&1; tail -12 gpurun_out/r2_t7.log
python tools/sia_stamps.py euroc > gpurun_out/r2_stamps7.log 2>&1; python tools/sia_stamps.py euroc fast >> gpurun_out/r2_stamps7.log 2>&1; grep -v amdgpu.ids gpurun_out/r2_stamps7.log
bash tools/profile_bench.sh r2_p5 --seqs 256 --steps 20 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
grep "svo::" gpurun_out/r2_p5/kernel_stats.csv | head -9
cut -c1-300 gpurun_out/r2_p5/bench.json; echo
timeout -k 10 500 python bench.py > gpurun_out/r2_bench4.json 2> gpurun_out/r2_bench4.err; tail -3 gpurun_out/r2_bench4.err; cut -c1-2600 gpurun_out/r2_bench4.json; echo
GPU_MAX_HW_QUEUES=8 SVO_GROUPS=6 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r2_bench4_q8g6.json 2> gpurun_out/r2_bench4_q8g6.err; cut -c1-200 gpurun_out/r2_bench4_q8g6.json; echo
GPU_MAX_HW_QUEUES=8 SVO_GROUPS=4 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r2_bench4_q8g4.json 2> gpurun_out/r2_bench4_q8g4.err; cut -c1-200 gpurun_out/r2_bench4_q8g4.json; echo
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --copy-input > gpurun_out/r2_bench4_copy.json 2> gpurun_out/r2_bench4_copy.err; cut -c1-200 gpurun_out/r2_bench4_copy.json; echo
