#!/bin/bash
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r2_t27.log 2>&1; tail -3 gpurun_out/r2_t27.log
B="--no-cpu-baseline --no-extras --repeats 1"
echo "== new pyramid"; timeout -k 10 300 python bench.py $B 2>/dev/null | cut -c1-120
echo "== again"; timeout -k 10 300 python bench.py $B 2>/dev/null | cut -c1-120
echo "== hd 96"; timeout -k 10 300 python bench.py --config hd --seqs 96 --steps 20 --warmup 4 --repeats 1 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-120
