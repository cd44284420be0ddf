#!/bin/bash
R=$GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-extras --repeats 1"
echo "== base"; timeout -k 10 300 python bench.py $B 2>/dev/null | cut -c1-120
echo "== kltocc4"; SVO_HIP_LIB=$R/build_ab/libsvo_hip_kltocc4.so timeout -k 10 300 python bench.py $B 2>/dev/null | cut -c1-120
echo "== pyr stream q12"; SVO_PYR_STREAM=1 timeout -k 10 300 python bench.py $B 2>/dev/null | cut -c1-120
echo "== pyr stream q20"; GPU_MAX_HW_QUEUES=20 SVO_PYR_STREAM=1 timeout -k 10 300 python bench.py $B 2>/dev/null | cut -c1-120
echo "== base again"; timeout -k 10 300 python bench.py $B 2>/dev/null | cut -c1-120
echo "== pyr stream tests"; SVO_PYR_STREAM=1 timeout -k 10 300 python -m pytest tests/test_tracker_gpu.py -m gpu -q -x -p no:cacheprovider -k "oracle or groups or batch_equals" 2>&1 | tail -1
