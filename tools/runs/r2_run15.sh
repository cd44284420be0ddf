#!/bin/bash
# scratch GPU-box script of round 2: alignment kernel with a small LDS footprint (image / records from L2)
R=$GRAFT_REPO_ROOT
for v in "SVO_SIA_MODE=1" "SVO_SIA_MODE=2" "SVO_SIA_MODE=2 SVO_SIA_WAVES=1" "SVO_SIA_MODE=2 SVO_HIP_LIB=$R/build_ab/libsvo_hip_stg16.so" "SVO_SIA_MODE=2 SVO_HIP_LIB=$R/build_ab/libsvo_hip_stg32.so" "SVO_SIA_MODE=2 SVO_SIA_WAVES=1 SVO_HIP_LIB=$R/build_ab/libsvo_hip_stg16.so" "SVO_SIA_MODE=0 SVO_SIA_WAVES=1"; do
  echo "== plain bench: $v"
  env $v timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --repeats 1 2>/dev/null | cut -c1-120
done
echo "== q8 g6 mode 2 stg16"; GPU_MAX_HW_QUEUES=8 SVO_GROUPS=6 SVO_SIA_MODE=2 SVO_HIP_LIB=$R/build_ab/libsvo_hip_stg16.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --repeats 1 2>/dev/null | cut -c1-120
echo "== g4 mode 2 stg16 1024"; SVO_GROUPS=4 SVO_SIA_MODE=2 SVO_HIP_LIB=$R/build_ab/libsvo_hip_stg16.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --repeats 1 --seqs 1024 2>/dev/null | cut -c1-120
A="--steps 10 --repeats 1 --no-cpu-baseline --no-extras"
echo "== timeline mode 2 stg16"; SVO_SIA_MODE=2 SVO_HIP_LIB=$R/build_ab/libsvo_hip_stg16.so bash tools/profile_timeline.sh r2_tl5_m2 $A
