#!/bin/bash
# the round's bench lines: default command with every leg, and the 1920x1080 case (C3) with its kernel stats
/usr/bin/time -v -o gpurun_out/r02_bench_default.time timeout -k 10 900 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; tail -2 gpurun_out/r02_bench_default.err; grep -E "Elapsed|Maximum resident" gpurun_out/r02_bench_default.time; cut -c1-300 gpurun_out/r02_bench_default.json; echo
bash tools/profile_bench.sh r02_c3 --config hd --seqs 48 --steps 20 --warmup 4 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1; head -12 gpurun_out/r02_c3/kernel_stats.csv | cut -c1-140
timeout -k 10 300 python bench.py --config hd --seqs 96 --steps 20 --warmup 4 --repeats 3 --no-cpu-baseline --no-extras > gpurun_out/r02_c3_bench.json 2> gpurun_out/r02_c3_bench.err; cut -c1-200 gpurun_out/r02_c3_bench.json; echo
