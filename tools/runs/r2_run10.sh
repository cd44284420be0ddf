#!/bin/bash
# scratch GPU-box script of round 2: overlap of the sequence groups, small-LDS alignment kernel
A="--steps 10 --repeats 1 --no-cpu-baseline --no-extras"
export SVO_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/libsvo_hip_stg16.so
export SVO_SIA_MODE=1
echo "== g1 mode1 stg16"; SVO_GROUPS=1 bash tools/profile_timeline.sh r2_tl2_g1 $A
echo "== g3 mode1 stg16"; bash tools/profile_timeline.sh r2_tl2_g3 $A
echo "== q8 g6 mode1 stg16"; GPU_MAX_HW_QUEUES=8 SVO_GROUPS=6 bash tools/profile_timeline.sh r2_tl2_g6 $A
unset SVO_HIP_LIB
echo "== g3 fast solver"; bash tools/profile_timeline.sh r2_tl2_fast $A --fast
