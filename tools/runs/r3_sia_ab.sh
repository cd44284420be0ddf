#!/bin/bash
# alignment kernel variants, one sequence group of 256 alone on the GPU + cycle stamps of a lone sequence
A="--no-extras --repeats 1 --seqs 256 --loops 16 --steps 30"
export SVO_GROUPS=1
bash tools/ab_bench.sh "$A" stereo-svo-slam_amd/csrc/libsvo_hip.so build_ab/libsvo_hip_accu2.so
unset SVO_GROUPS
python tools/sia_stamps.py euroc
