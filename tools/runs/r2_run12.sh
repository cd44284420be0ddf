#!/bin/bash
# scratch GPU-box script of round 2: four keypoints per accumulate trip; does a one-lane solve change co-resident speed?
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py tests/test_golden_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r2_t12.log 2>&1; tail -2 gpurun_out/r2_t12.log
python tools/sia_stamps.py euroc > gpurun_out/r2_stamps12.log 2>&1; grep -v amdgpu.ids gpurun_out/r2_stamps12.log | head -12
A="--steps 10 --repeats 1 --no-cpu-baseline --no-extras"
echo "== g1 mode0 base"; SVO_GROUPS=1 bash tools/profile_timeline.sh r2_tl4_a $A | grep -E "value|sia_gn_kernel<1|reproj_gn_kernel<1|klt"
echo "== g1 mode0 onelane"; SVO_HIP_LIB=$R/build_ab/libsvo_hip_onelane.so SVO_GROUPS=1 bash tools/profile_timeline.sh r2_tl4_b $A | grep -E "value|sia_gn_kernel<1|reproj_gn_kernel<1|klt"
echo "== g1 mode1 stg16 base"; SVO_SIA_MODE=1 SVO_HIP_LIB=$R/build_ab/libsvo_hip_stg16.so SVO_GROUPS=1 bash tools/profile_timeline.sh r2_tl4_c $A | grep -E "value|sia_gn_kernel<1|reproj_gn_kernel<1|klt"
echo "== g1 mode1 stg16 onelane"; SVO_SIA_MODE=1 SVO_HIP_LIB=$R/build_ab/libsvo_hip_onelane16.so SVO_GROUPS=1 bash tools/profile_timeline.sh r2_tl4_d $A | grep -E "value|sia_gn_kernel<1|reproj_gn_kernel<1|klt"
for v in "SVO_SIA_MODE=0" "SVO_SIA_MODE=0 SVO_HIP_LIB=$R/build_ab/libsvo_hip_onelane.so" "SVO_SIA_MODE=1 SVO_HIP_LIB=$R/build_ab/libsvo_hip_onelane16.so" "SVO_SIA_MODE=1 SVO_HIP_LIB=$R/build_ab/libsvo_hip_onelane.so"; do
  echo "== plain bench: $v"
  env $v timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --repeats 1 2>/dev/null | cut -c1-120
done
