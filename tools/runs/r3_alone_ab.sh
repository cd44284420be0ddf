#!/bin/bash
# one sequence group of 256 alone on the GPU: stage times of library builds
A="--no-extras --repeats 1 --seqs 256 --loops 16 --steps 30"
export SVO_GROUPS=1
bash tools/ab_bench.sh "$A" "$@"
