#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace CSV: how the kernels of the sequence groups overlap.

  python tools/timeline.py <kernel_trace.csv> [skip_fraction]

Prints, for the second half of the trace (the timed steps): the union of busy time, the sum of kernel
durations, per queue the busy time and the idle gaps between consecutive kernels, per kernel name the
average duration and the average number of OTHER kernels running at the same time, and how much of
the wall time had 0, 1, 2, ... kernels in flight.
"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("svo::", "")
    cut = name.find("(")
    return (name if cut < 0 else name[:cut])[:44]


def main():
    path = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"),
                         short(r["Kernel_Name"])))
    rows.sort()
    # the window: the last (1 - skip) of the alignment kernel's dispatches (the timed steps), not of
    # the wall time (set-up and rendering come first)
    sia = [r for r in rows if "sia_gn_kernel" in r[3]]
    if sia:
        t_lo = sia[int(skip * len(sia))][0]
        t_hi = sia[-1][1]
        rows = [r for r in rows if r[0] >= t_lo and r[1] <= t_hi]
    else:
        t_lo = rows[0][0] + skip * (rows[-1][1] - rows[0][0])
        rows = [r for r in rows if r[0] >= t_lo]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    wall = t1 - t0
    print(f"{len(rows)} dispatches over {wall / 1e6:.2f} ms")
    # concurrency profile
    ev = []
    for s, e, q, n in rows:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    level_time = defaultdict(int)
    lvl, last = 0, t0
    for t, d in ev:
        level_time[lvl] += t - last
        last = t
        lvl += d
    busy = wall - level_time[0]
    print(f"busy (>=1 kernel): {busy / wall:.3f} of the wall time; sum of durations / wall = "
          f"{sum(e - s for s, e, _, _ in rows) / wall:.2f}")
    print("kernels in flight: " + "  ".join(f"{k}:{v / wall:.3f}" for k, v in sorted(level_time.items())))
    # per queue
    byq = defaultdict(list)
    for r in rows:
        byq[r[2]].append(r)
    for q, rs in sorted(byq.items()):
        b = sum(e - s for s, e, _, _ in rs)
        gaps = [max(0, rs[i + 1][0] - rs[i][1]) for i in range(len(rs) - 1)]
        big = sorted(gaps)[-5:]
        print(f"queue {q}: {len(rs)} kernels, busy {b / wall:.3f}, mean gap {sum(gaps) / max(1, len(gaps)) / 1e3:.1f} us, "
              f"largest gaps {[round(g / 1e3) for g in big]} us")
    # per kernel: duration and overlap
    starts = sorted((s, e) for s, e, _, _ in rows)
    byn = defaultdict(list)
    for s, e, q, n in rows:
        ov = 0
        for s2, e2 in starts:
            if s2 >= e:
                break
            if e2 > s:
                ov += min(e, e2) - max(s, s2)
        byn[n].append((e - s, (ov - (e - s)) / max(1, e - s)))
    print(f"{'kernel':46s} {'calls':>6s} {'avg us':>9s} {'min us':>9s} {'max us':>9s} {'others in flight':>17s}")
    for n, v in sorted(byn.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
        d = [x for x, _ in v]
        print(f"{n:46s} {len(v):6d} {sum(d) / len(d) / 1e3:9.1f} {min(d) / 1e3:9.1f} {max(d) / 1e3:9.1f} "
              f"{sum(o for _, o in v) / len(v):17.2f}")


if __name__ == "__main__":
    main()
