#!/bin/bash
# SQ counters of the alignment kernel in one configuration (two passes of 8 counters), condensed to
# means per dispatch. usage: pmc_sia.sh <tag> [bench args]   (environment variables are inherited)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
A="--steps 6 --warmup 4 --repeats 1 --no-cpu-baseline --no-extras $@"
rm -rf /tmp/pmc_$TAG
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-include-regex "sia_gn" --output-format csv -d /tmp/pmc_$TAG/a -- python3 bench.py $A > "$OUT/a.json" 2> "$OUT/a.err"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-include-regex "sia_gn" --output-format csv -d /tmp/pmc_$TAG/b -- python3 bench.py $A > "$OUT/b.json" 2> "$OUT/b.err"
python3 - "$TAG" <<'PY' > "$OUT/pmc.txt"
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(f"/tmp/pmc_{tag}/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void svo::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k, "dispatches", max(len(v) for v in agg[k].values()))
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:26s} {sum(v) / len(v):16.0f}")
PY
cat "$OUT/pmc.txt"
