#!/bin/bash
A="--steps 10 --repeats 1 --no-cpu-baseline --no-extras --seqs 768"
SVO_GROUPS=1 bash tools/profile_timeline.sh r02_tl_alone $A | tail -20
