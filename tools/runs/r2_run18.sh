#!/bin/bash
# scratch GPU-box script of round 2: KLT without the scratch copy, 4-wave cap; smaller workgroups for SSD / KLT
R=$GRAFT_REPO_ROOT
P="tests/test_parity_gpu.py tests/test_golden_gpu.py"
timeout -k 10 300 python -m pytest $P -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1
for v in base ssd64 ssd128 klt64 klt64ssd64; do
  L=$R/build_ab/libsvo_hip_$v.so; [ $v = base ] && L=$R/stereo-svo-slam_amd/csrc/libsvo_hip.so
  echo "== $v"
  [ $v != base ] && SVO_HIP_LIB=$L timeout -k 10 300 python -m pytest $P -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1
  SVO_HIP_LIB=$L timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --repeats 1 2>/dev/null | cut -c1-120
done
echo "== base again"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --repeats 1 2>/dev/null | cut -c1-120
echo "== base fast"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --repeats 1 --fast 2>/dev/null | cut -c1-120
echo "== hd 48"; timeout -k 10 300 python bench.py --config hd --seqs 48 --steps 20 --warmup 4 --repeats 1 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-120
A="--steps 10 --repeats 1 --no-cpu-baseline --no-extras"
bash tools/profile_timeline.sh r2_tl7 $A | tail -18
