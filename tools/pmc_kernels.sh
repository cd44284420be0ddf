#!/bin/bash
# Collects PMC counters for our kernels only (one rocprofv3 pass per counter group) and
# prints per-kernel averages. Usage: pmc_kernels.sh <tag> "<counters>" [bench args...]
set -e
TAG=$1; CTRS=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$TAG
cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc $CTRS --kernel-include-regex "svo" --output-format csv -d /tmp/pmc_$TAG -- python3 bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
F=$(find /tmp/pmc_$TAG -name "*counter_collection.csv" | head -1)
python3 - "$F" "$OUT/pmc_summary.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    f.write("kernel,counter,dispatches,mean,max\n")
    for k in sorted(agg):
        for c in sorted(agg[k]):
            v = agg[k][c]
            f.write(f"{k},{c},{len(v)},{sum(v)/len(v):.1f},{max(v):.1f}\n")
print(open(sys.argv[2]).read())
PY
