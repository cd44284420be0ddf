#!/bin/bash
R=$GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-extras --repeats 1"
for v in "SVO_SIA_MODE=2 SVO_SIA_WAVES=1" "SVO_SIA_MODE=2 SVO_SIA_WAVES=1 SVO_HIP_LIB=$R/build_ab/libsvo_hip_stg32.so" "SVO_SIA_MODE=2 SVO_SIA_WAVES=1 SVO_HIP_LIB=$R/build_ab/libsvo_hip_stg16.so" "SVO_SIA_MODE=2 SVO_SIA_WAVES=2 SVO_HIP_LIB=$R/build_ab/libsvo_hip_stg32.so"; do
  echo "== $v"; env $v timeout -k 10 300 python bench.py $B 2>/dev/null | cut -c1-120
done
echo "== mode2 w1 2048/8 q12"; GPU_MAX_HW_QUEUES=12 SVO_SIA_MODE=2 SVO_SIA_WAVES=1 timeout -k 10 400 python bench.py $B --seqs 2048 2>/dev/null | cut -c1-120
echo "== mode2 w1 stg32 2048/8 q12"; GPU_MAX_HW_QUEUES=12 SVO_SIA_MODE=2 SVO_SIA_WAVES=1 SVO_HIP_LIB=$R/build_ab/libsvo_hip_stg32.so timeout -k 10 400 python bench.py $B --seqs 2048 2>/dev/null | cut -c1-120
