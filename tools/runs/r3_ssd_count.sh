#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  rm -rf /tmp/kc
  SVO_HIP_LIB=$GRAFT_REPO_ROOT/$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT --kernel-include-regex "ssd" --output-format csv -d /tmp/kc -- python3 tools/ssd_count.py > /tmp/kc.out 2>/tmp/kc.err
  python3 - "$lib" "$(find /tmp/kc -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])): agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = agg["SQ_WAVES"][-1]
print(sys.argv[1], {k: round(v[-1] / w, 1) for k, v in agg.items()}, "waves", w)
PY
done
