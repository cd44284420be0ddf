#!/bin/bash
# the round's committed numbers: default bench line, profile set, C3 run
S=$(date +%s)
timeout -k 10 400 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench rc=$? $(( $(date +%s) - S )) s"
cut -c1-400 gpurun_out/r03_bench_default.json; echo
bash tools/profile_r03.sh r03 --steps 40 > gpurun_out/r03_profile.log 2>&1; echo "profile rc=$?"
bash tools/runs/r3_hd.sh 2>&1 | tail -12 | cut -c1-200
