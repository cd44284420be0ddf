#!/bin/bash
# GPU timeline of the last steps of a short bench run: kernel + memory-copy trace from rocprofv3,
# condensed to "start offset, duration, name" lines (gpurun_out/<tag>/timeline.txt).
set -e
TAG=${1:-timeline}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl_$TAG
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_$TAG -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 - /tmp/tl_$TAG "$OUT/timeline.txt" <<'PY'
import csv, glob, sys
d, out = sys.argv[1], sys.argv[2]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") ))
ev.sort()
# last 3 steps: find the last three pyr_halfsample kernels
idx = [i for i, e in enumerate(ev) if "pyr_halfsample" in e[2]]
start = idx[-3] if len(idx) >= 3 else 0
t0 = ev[start][0]
with open(out, "w") as f:
    prev_end = t0
    for s, e, n in ev[start:]:
        f.write(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:8.1f}  {n}\n")
        prev_end = max(prev_end, e)
# utilisation over the listed window: union of busy intervals vs span, and per-kernel sums
span0, span1 = ev[start][0], max(e for _, e, _ in ev[start:])
busy, cur_s, cur_e = 0, None, None
for s_, e_, _ in ev[start:]:
    if cur_e is None or s_ > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
busy += cur_e - cur_s
import collections
per = collections.defaultdict(float)
for s_, e_, n_ in ev[start:]: per[n_] += (e_ - s_) / 1e3
with open(out, "a") as f:
    f.write(f"\nspan {(span1 - span0) / 1e3:.1f} us, some kernel or copy running {busy / 1e3:.1f} us ({100.0 * busy / (span1 - span0):.1f} %)\n")
    for n_, t_ in sorted(per.items(), key=lambda kv: -kv[1]): f.write(f"  {t_:10.1f} us  {n_}\n")
print(open(out).read()[-3000:])
PY
