#!/bin/bash
# Round-3 profile set of the DEFAULT bench command (without the CPU / extra legs):
#   1. rocprofv3 --kernel-trace --stats           -> kernel_stats.csv, bench_under_rocprof.json
#   2. rocprofv3 --pmc <SQ counters> (own pass)   -> pmc_sq.csv
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each: they do not fit together)
# and tools/make_pmc_json.py condenses them into pmc.json / traffic.json (read by bench.py from
# profiles/r03_*). Usage on the GPU box: bash tools/profile_r03.sh <tag> [bench args]
set -e
TAG=${1:-r03}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
ARGS="--no-cpu-baseline --no-extras --repeats 1 $@"
rm -rf /tmp/rp_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG/kt -- python3 bench.py $ARGS > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err"
find /tmp/rp_$TAG/kt -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
echo "kernel stats done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-include-regex "svo" --output-format csv -d /tmp/rp_$TAG/sq -- python3 bench.py $ARGS > "$OUT/bench_pmc_sq.json" 2> "$OUT/sq.err"
cp "$(find /tmp/rp_$TAG/sq -name '*counter_collection.csv' | head -1)" /tmp/rp_$TAG/sq.csv
echo "SQ pass done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "svo" --output-format csv -d /tmp/rp_$TAG/$C -- python3 bench.py $ARGS > "$OUT/bench_pmc_$C.json" 2> "$OUT/$C.err"
  cp "$(find /tmp/rp_$TAG/$C -name '*counter_collection.csv' | head -1)" /tmp/rp_$TAG/$C.csv
  echo "$C pass done"
done
python3 tools/make_pmc_json.py /tmp/rp_$TAG "$OUT"
# kernel times alone on the GPU: one sequence group of 256
SVO_GROUPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG/kt1 -- python3 bench.py --no-cpu-baseline --no-extras --repeats 1 --seqs 256 --loops 16 --steps 40 > /dev/null 2> "$OUT/kt1.err"
find /tmp/rp_$TAG/kt1 -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_1group_256seq.csv" \;
echo "alone pass done"
# diagnostic passes (what the co-running kernels compete for); a counter the device does not have fails its own pass only
for SET in "SQ_BUSY_CYCLES SQ_LEVEL_WAVES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  N=$(echo $SET | cut -d" " -f1)
  if rocprofv3 --pmc $SET --kernel-include-regex "svo" --output-format csv -d /tmp/rp_$TAG/d_$N -- python3 bench.py $ARGS > /dev/null 2> "$OUT/d_$N.err"; then
    python3 - "$(find /tmp/rp_$TAG/d_$N -name '*counter_collection.csv' | head -1)" "$OUT/diag_$N.csv" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("svo::", "").split("<")[0]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    f.write("kernel,counter,dispatches,mean_per_dispatch\n")
    for k in sorted(agg):
        for c in sorted(agg[k]):
            v = agg[k][c]
            f.write(f"{k},{c},{len(v)},{sum(v)/len(v):.1f}\n")
print(open(sys.argv[2]).read())
PY
  else
    echo "diagnostic pass $N failed"; tail -3 "$OUT/d_$N.err"
  fi
done
