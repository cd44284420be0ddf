#!/bin/bash
# the pyramid kernel alone on the GPU (one sequence group of 256), per library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  rm -rf /tmp/pa
  SVO_HIP_LIB=$GRAFT_REPO_ROOT/$lib SVO_GROUPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -- python3 bench.py --no-cpu-baseline --no-extras --repeats 1 --seqs 256 --loops 16 --steps 30 > /dev/null 2>&1
  echo "$lib: $(grep pyr_stream $(find /tmp/pa -name '*kernel_stats.csv' | head -1) | cut -d, -f2-4,6,7)"
done
