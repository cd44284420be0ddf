#!/bin/bash
# scratch GPU-box script of round 2: how the sequence groups overlap
A="--steps 10 --repeats 1 --no-cpu-baseline --no-extras"
echo "== default (mode 0, stg 64)"; bash tools/profile_timeline.sh r2_tl_m0 $A
echo "== mode 1"; SVO_SIA_MODE=1 bash tools/profile_timeline.sh r2_tl_m1 $A
echo "== 1 group"; SVO_GROUPS=1 bash tools/profile_timeline.sh r2_tl_g1 $A
echo "== occ2"; SVO_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/libsvo_hip_occ2.so bash tools/profile_timeline.sh r2_tl_occ2 $A
for v in "SVO_SIA_MODE=0" "SVO_SIA_MODE=1" "SVO_SIA_MODE=1 SVO_SIA_WAVES=2" "SVO_SIA_MODE=0 SVO_SIA_WAVES=2" "SVO_SIA_MODE=1 SVO_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/libsvo_hip_occ2.so" "SVO_SIA_MODE=1 SVO_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/libsvo_hip_stg16.so"; do
  echo "== plain bench: $v"
  env $v timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --repeats 1 2>/dev/null | cut -c1-120
done
