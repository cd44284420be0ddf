#!/bin/bash
# HBM traffic of our kernels for the default bench command: FETCH_SIZE and WRITE_SIZE in
# SEPARATE rocprofv3 --pmc passes (they do not fit one pass on gfx950), our kernels only.
# Writes gpurun_out/<tag>/{fetch.csv,write.csv,traffic.json}.
set -e
TAG=${1:-traffic}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --pmc $C --kernel-include-regex "svo" --output-format csv -d /tmp/pmc_$C -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench_$C.json" 2> "$OUT/bench_$C.err"
  cp "$(find /tmp/pmc_$C -name '*counter_collection.csv' | head -1)" /tmp/pmc_$C.csv
done
python3 - "$OUT" "$@" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
args = sys.argv[2:]
bj = json.loads(open(f"{out}/bench_FETCH_SIZE.json").read().strip().splitlines()[-1])
seqs, groups = bj["config"]["sequences_per_gpu"], bj["config"]["sequence_groups"]
cfg = args[args.index("--config") + 1] if "--config" in args else "euroc"
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"/tmp/pmc_{c}.csv")):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("svo::", "").split("<")[0]
        agg[name].append(float(r["Counter_Value"]))
    with open(f"{out}/{c.lower()}.csv", "w") as f:
        f.write("kernel,dispatches,mean_KB,max_KB\n")
        for k, v in sorted(agg.items()):
            # the first dispatch of each kernel belongs to warm-up / first frame; keep all, report mean
            f.write(f"{k},{len(v)},{sum(v)/len(v):.1f},{max(v):.1f}\n")
            res[k][c] = sum(v) / len(v)
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB. gfx950: FETCH_SIZE counts 128-B requests of wide
# coalesced streams as 64 B (x2 there); these kernels read bytes/dwords per lane, an access shape the
# guide leaves uncalibrated, so the raw value is kept and the x2 upper bound is stored beside it.
traffic = {k: (v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 for k, v in res.items()}
upper = {k: (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 for k, v in res.items()}
json.dump({"seqs": seqs, "groups": groups, "config": cfg, "bytes_per_launch": traffic, "bytes_per_launch_fetch_x2": upper,
           "note": "mean over all dispatches of the run; FETCH_SIZE and WRITE_SIZE from separate --pmc passes"},
          open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(traffic, indent=1))
PY
