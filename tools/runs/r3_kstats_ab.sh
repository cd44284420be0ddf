#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace --stats) of library builds, one sequence group of 256 alone
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SVO_GROUPS=1
for lib in "$@"; do
  rm -rf /tmp/ks
  SVO_HIP_LIB=$GRAFT_REPO_ROOT/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 bench.py --no-cpu-baseline --no-extras --repeats 1 --seqs 256 --loops 16 --steps 40 > /tmp/ks.json 2> /tmp/ks.err || { tail -3 /tmp/ks.err; exit 1; }
  python3 - "$lib" "$(find /tmp/ks -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[2])):
    n = r["Name"].replace("void ", "").replace("svo::", "").split("(")[0]
    if n.split("<")[0] in ("klt_track_kernel", "ssd_disparity_kernel", "kf_detect_kernel", "pyr_stream_kernel", "sia_prep_kernel", "sia_gn_kernel", "reproj_gn_kernel"):
        out.append("%s %.1f us x %s" % (n.split("_kernel")[0], float(r["AverageNs"]) / 1e3, r["Calls"]))
print(sys.argv[1], "|", " | ".join(out), flush=True)
PY
done
