#!/bin/bash
# bench under rocprofv3 --kernel-trace; keeps the stats CSV and the overlap summary of tools/timeline.py
# usage: profile_timeline.sh <tag> [bench args...]   (environment variables are inherited)
TAG=${1:-tl}
shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rm -rf /tmp/rocprof_$TAG
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rocprof_$TAG -- python3 bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
find /tmp/rocprof_$TAG -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
TR=$(find /tmp/rocprof_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py "$TR" 0.6 > "$OUT/timeline.txt" 2>&1
cut -c1-160 "$OUT/bench.json"; echo
cat "$OUT/timeline.txt"
