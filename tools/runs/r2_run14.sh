#!/bin/bash
# scratch GPU-box script of round 2: tests, the profile set of the default bench, the 1920x1080 case
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r2_t14.log 2>&1; tail -3 gpurun_out/r2_t14.log
bash tools/profile_r02.sh r02 > gpurun_out/r02_profile.log 2>&1; tail -5 gpurun_out/r02_profile.log
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r2_bench14.json 2> gpurun_out/r2_bench14.err; cut -c1-200 gpurun_out/r2_bench14.json; echo
timeout -k 10 300 python bench.py --config hd --seqs 48 --steps 20 --warmup 4 --repeats 3 --no-cpu-baseline --no-extras > gpurun_out/r2_bench14_hd.json 2> gpurun_out/r2_bench14_hd.err; tail -2 gpurun_out/r2_bench14_hd.err; cut -c1-300 gpurun_out/r2_bench14_hd.json; echo
