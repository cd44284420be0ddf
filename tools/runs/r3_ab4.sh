#!/bin/bash
A="--no-extras --repeats 1 --steps 60"
bash tools/ab_bench.sh "$A" "$@"
