#!/bin/bash
# Runs the default bench under rocprofv3 --kernel-trace --stats and keeps only the
# summaries (the per-dispatch trace is tens of MB): gpurun_out/<tag>/{kernel_stats.csv,bench.json}
set -e
TAG=${1:-prof}
shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rocprof_$TAG
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rocprof_$TAG -- python3 bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
find /tmp/rocprof_$TAG -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find /tmp/rocprof_$TAG -name "*domain_stats.csv" -exec cp {} "$OUT/domain_stats.csv" \;
head -20 "$OUT/kernel_stats.csv"
