#!/bin/bash
# the round's bench line: default command with every leg
S=$(date +%s)
timeout -k 10 1000 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; tail -2 gpurun_out/r02_bench_default.err
echo "wall seconds: $(( $(date +%s) - S ))" | tee gpurun_out/r02_bench_default.time
cut -c1-300 gpurun_out/r02_bench_default.json; echo
