#!/bin/bash
# the round's profile set of the default bench command
bash tools/profile_r02.sh r02 > gpurun_out/r02_profile.log 2>&1; tail -4 gpurun_out/r02_profile.log
A="--steps 10 --repeats 1 --no-cpu-baseline --no-extras"
bash tools/profile_timeline.sh r02_tl_default $A | tail -22
