#!/bin/bash
timeout -k 10 400 python -m pytest tests/test_tracker_gpu.py -m gpu -q -x -p no:cacheprovider -k "batched or groups or unequal or two_ranks or batch_equals" 2>&1 | tail -2
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py tests/test_golden_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1
B="--no-cpu-baseline --no-extras --repeats 1"
echo "== default 2048 q12"; timeout -k 10 400 python bench.py $B 2>/dev/null | cut -c1-120
echo "== 2560/10 q12"; timeout -k 10 400 python bench.py $B --seqs 2560 2>/dev/null | cut -c1-120
echo "== 3072/12 q16"; GPU_MAX_HW_QUEUES=16 timeout -k 10 500 python bench.py $B --seqs 3072 2>/dev/null | cut -c1-120
echo "== 2048/8 q12 fast"; timeout -k 10 400 python bench.py $B --fast 2>/dev/null | cut -c1-120
echo "== 1536/6"; timeout -k 10 400 python bench.py $B --seqs 1536 2>/dev/null | cut -c1-120
