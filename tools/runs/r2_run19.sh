#!/bin/bash
# scratch GPU-box script of round 2: more sequence groups / sequences in flight with 8 hardware queues
R=$GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-extras --repeats 1"
echo "== base 768/3"; timeout -k 10 200 python bench.py $B 2>/dev/null | cut -c1-120
export GPU_MAX_HW_QUEUES=8
echo "== q8 768/3"; timeout -k 10 200 python bench.py $B 2>/dev/null | cut -c1-120
echo "== q8 768/4"; SVO_GROUPS=4 timeout -k 10 200 python bench.py $B 2>/dev/null | cut -c1-120
echo "== q8 768/6"; SVO_GROUPS=6 timeout -k 10 200 python bench.py $B 2>/dev/null | cut -c1-120
echo "== q8 1024/4"; SVO_GROUPS=4 timeout -k 10 300 python bench.py $B --seqs 1024 2>/dev/null | cut -c1-120
echo "== q8 1536/6"; SVO_GROUPS=6 timeout -k 10 300 python bench.py $B --seqs 1536 2>/dev/null | cut -c1-120
echo "== q8 1280/5"; SVO_GROUPS=5 timeout -k 10 300 python bench.py $B --seqs 1280 2>/dev/null | cut -c1-120
echo "== q8 768/6 klt64"; SVO_HIP_LIB=$R/build_ab/libsvo_hip_klt64.so SVO_GROUPS=6 timeout -k 10 200 python bench.py $B 2>/dev/null | cut -c1-120
echo "== q8 1536/6 klt64"; SVO_HIP_LIB=$R/build_ab/libsvo_hip_klt64.so SVO_GROUPS=6 timeout -k 10 300 python bench.py $B --seqs 1536 2>/dev/null | cut -c1-120
