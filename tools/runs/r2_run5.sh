#!/bin/bash
# scratch GPU-box script of round 2
timeout -k 10 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r2_t5.log 2>&1; tail -12 gpurun_out/r2_t5.log
python tools/sia_stamps.py euroc > gpurun_out/r2_stamps5.log 2>&1; python tools/sia_stamps.py euroc exact >> gpurun_out/r2_stamps5.log 2>&1; grep -v amdgpu.ids gpurun_out/r2_stamps5.log
bash tools/profile_bench.sh r2_p3 --seqs 256 --steps 20 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
bash tools/profile_bench.sh r2_p3x --seqs 256 --steps 20 --repeats 1 --no-cpu-baseline --no-extras --fast > /dev/null 2>&1
grep "svo::" gpurun_out/r2_p3/kernel_stats.csv | head -9
grep "svo::" gpurun_out/r2_p3x/kernel_stats.csv | head -5
cut -c1-300 gpurun_out/r2_p3/bench.json; echo; cut -c1-300 gpurun_out/r2_p3x/bench.json; echo
timeout -k 10 500 python bench.py > gpurun_out/r2_bench2.json 2> gpurun_out/r2_bench2.err; tail -3 gpurun_out/r2_bench2.err; cut -c1-2500 gpurun_out/r2_bench2.json
timeout -k 10 500 python bench.py --exact --no-cpu-baseline > gpurun_out/r2_bench2x.json 2> gpurun_out/r2_bench2x.err; tail -3 gpurun_out/r2_bench2x.err; cut -c1-1500 gpurun_out/r2_bench2x.json
