#!/bin/bash
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r2_t20.log 2>&1; tail -3 gpurun_out/r2_t20.log
B="--no-cpu-baseline --no-extras --repeats 1"
echo "== default (1536, q8)"; timeout -k 10 300 python bench.py $B 2>/dev/null | cut -c1-120
echo "== 768"; timeout -k 10 300 python bench.py $B --seqs 768 2>/dev/null | cut -c1-120
echo "== 1792/7"; timeout -k 10 300 python bench.py $B --seqs 1792 2>/dev/null | cut -c1-120
echo "== q12 2048/8"; GPU_MAX_HW_QUEUES=12 timeout -k 10 400 python bench.py $B --seqs 2048 2>/dev/null | cut -c1-120
echo "== q12 2560/10"; GPU_MAX_HW_QUEUES=12 timeout -k 10 400 python bench.py $B --seqs 2560 2>/dev/null | cut -c1-120
echo "== q8 1536 fast"; timeout -k 10 300 python bench.py $B --fast 2>/dev/null | cut -c1-120
