#!/bin/bash
# scratch GPU-box script of round 2: LDS-resident SVD
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_golden_gpu.py tests/test_tracker_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r2_t11.log 2>&1; tail -4 gpurun_out/r2_t11.log
python tools/sia_stamps.py euroc > gpurun_out/r2_stamps11.log 2>&1; grep -v amdgpu.ids gpurun_out/r2_stamps11.log | head -12
A="--steps 10 --repeats 1 --no-cpu-baseline --no-extras"
echo "== g3 mode0"; bash tools/profile_timeline.sh r2_tl3_g3 $A
echo "== g1 mode0"; SVO_GROUPS=1 bash tools/profile_timeline.sh r2_tl3_g1 $A
echo "== g1 mode1"; SVO_SIA_MODE=1 SVO_GROUPS=1 bash tools/profile_timeline.sh r2_tl3_g1m1 $A
for v in "SVO_SIA_MODE=0" "SVO_SIA_MODE=1" "SVO_SIA_MODE=1 SVO_SIA_WAVES=2" "SVO_SIA_MODE=0 SVO_SIA_WAVES=2"; do
  echo "== plain bench: $v"
  env $v timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --repeats 1 2>/dev/null | cut -c1-120
done
