#!/bin/bash
# scratch GPU-box script of round 2
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r2_t8.log 2>&1; tail -5 gpurun_out/r2_t8.log
python tools/sia_stamps.py euroc > gpurun_out/r2_stamps8.log 2>&1; grep -v amdgpu.ids gpurun_out/r2_stamps8.log
bash tools/profile_bench.sh r2_p6 --seqs 256 --steps 20 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
grep "svo::" gpurun_out/r2_p6/kernel_stats.csv | head -9
cut -c1-300 gpurun_out/r2_p6/bench.json; echo
timeout -k 10 500 python bench.py --no-cpu-baseline > gpurun_out/r2_bench5.json 2> gpurun_out/r2_bench5.err; tail -3 gpurun_out/r2_bench5.err; cut -c1-2600 gpurun_out/r2_bench5.json; echo
GPU_MAX_HW_QUEUES=8 SVO_GROUPS=6 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r2_bench5_q8g6.json 2> gpurun_out/r2_bench5_q8g6.err; cut -c1-200 gpurun_out/r2_bench5_q8g6.json; echo
SVO_GROUPS=2 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r2_bench5_g2.json 2> gpurun_out/r2_bench5_g2.err; cut -c1-200 gpurun_out/r2_bench5_g2.json; echo
SVO_GROUPS=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r2_bench5_g1.json 2> gpurun_out/r2_bench5_g1.err; cut -c1-200 gpurun_out/r2_bench5_g1.json; echo
