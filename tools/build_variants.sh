#!/bin/bash
# Diagnostic builds of libsvo_hip into build_ab/ (git-ignored; travels to the GPU box):
#   build_variants.sh stamps            -> build_ab/libsvo_hip_stamps.so  (-DSVO_SIA_STAMPS)
#   build_variants.sh <tag> <flags...>  -> build_ab/libsvo_hip_<tag>.so   (extra hipcc flags)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/stereo-svo-slam_amd/csrc
TAG=$1; shift
FLAGS="$@"
[ "$TAG" = stamps ] && FLAGS="-DSVO_SIA_STAMPS $FLAGS"
mkdir -p $ROOT/build_ab/$TAG
for f in svo_capi svo_ctx pyramid sia klt reproj depth keyframe; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 $FLAGS -c $CSRC/$f.hip -o $ROOT/build_ab/$TAG/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $ROOT/build_ab/libsvo_hip_$TAG.so $ROOT/build_ab/$TAG/*.o
ls -la $ROOT/build_ab/libsvo_hip_$TAG.so
