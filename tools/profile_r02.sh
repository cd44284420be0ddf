#!/bin/bash
# Round-2 profile set of the DEFAULT bench command (without the CPU / extra legs):
#   1. rocprofv3 --kernel-trace --stats           -> kernel_stats.csv, bench_under_rocprof.json
#   2. rocprofv3 --pmc <SQ counters> (own pass)   -> pmc_sq.csv
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each: they do not fit together)
# and tools/make_pmc_json.py condenses them into pmc.json / traffic.json (read by bench.py from
# profiles/r02_*). Usage on the GPU box: bash tools/profile_r02.sh <tag> [bench args]
set -e
TAG=${1:-r02}; shift || true
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
