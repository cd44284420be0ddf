"""Condenses the rocprofv3 outputs of tools/profile_r02.sh into the files bench.py reads:
pmc.json (per kernel: the roof that binds it and the fraction of it that is reached) and
traffic.json (HBM bytes per launch: FETCH_SIZE corrected x2 as the guide prescribes for gfx950,
+ WRITE_SIZE). Usage: make_pmc_json.py <dir with sq.csv FETCH_SIZE.csv WRITE_SIZE.csv> <out dir>
(or, from the condensed files: make_pmc_json.py <dir with pmc_sq.csv kernel_stats.csv pmc.json> <out dir> reclassify)"""
import collections
import csv
import json
import sys

src, out = sys.argv[1], sys.argv[2]
N_SIMD = 256 * 4
CLK = 2.4e9                      # max shader clock, MI355X_MICROARCH.md
VALU_CYCLES = 2                  # a wave64 VALU instruction occupies a SIMD-32 for 2 cycles


def short(name):
    return name.split("(")[0].replace("void ", "").replace("svo::", "").split("<")[0]


def collect(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def weighted_durations(stats_csv):
    """mean dispatch time per kernel (template variants merged, weighted by their call counts)"""
    tot, calls = collections.defaultdict(float), collections.defaultdict(float)
    for r in csv.DictReader(open(stats_csv)):
        if "svo::" in r["Name"]:
            tot[short(r["Name"])] += float(r["TotalDurationNs"]) * 1e-9
            calls[short(r["Name"])] += float(r["Calls"])
    return {k: tot[k] / calls[k] for k in tot}


def call_counts(stats_csv):
    calls = collections.defaultdict(float)
    for r in csv.DictReader(open(stats_csv)):
        if "svo::" in r["Name"]:
            calls[short(r["Name"])] += float(r["Calls"])
    return calls


def valu_per_frame(sq, calls, seqs_per_launch):
    """VALU wave-instructions of one tracked frame of one sequence: every kernel's mean per dispatch x its
    dispatches, over the alignment launches (one per group and step) x the sequences of a launch."""
    steps = calls.get("sia_gn_kernel", 0.0)
    if steps <= 0:
        return None
    total = sum(c.get("SQ_INSTS_VALU", 0.0) * calls.get(k, 0.0) for k, c in sq.items())
    return total / (steps * seqs_per_launch)


def issue_activity(sq, calls, groups, ms_per_step):
    """Sum over all kernels of SQ_ACTIVE_INST_ANY (quad-cycles in which a wave issues an instruction of any kind)
    of one step x 4 / (step time x clock x SIMDs): how much of the chip's instruction issue the step uses,
    and every kernel's share of it."""
    steps = calls.get("sia_gn_kernel", 0.0) / max(groups, 1)
    if steps <= 0 or ms_per_step <= 0:
        return None
    per = {k: c.get("SQ_ACTIVE_INST_ANY", 0.0) * calls.get(k, 0.0) / steps for k, c in sq.items() if k in calls}
    tot = sum(per.values())
    return {"per_simd_cycle": tot * 4 / (ms_per_step * 1e-3 * CLK * N_SIMD),
            "share": {k: v / tot for k, v in sorted(per.items(), key=lambda kv: -kv[1]) if tot > 0 and v / tot >= 0.005},
            "ms_per_step": ms_per_step}


SERIAL = ("sia_gn_kernel", "reproj_gn_kernel")     # one or a few waves per sequence, a serial chain


def classify(sq, dur):
    kernels = {}
    for k, c in sq.items():
        if k not in dur or "SQ_WAVE_CYCLES" not in c:
            continue
        wc = max(c["SQ_WAVE_CYCLES"], 1.0)
        active, wait, issue_stall = c.get("SQ_ACTIVE_INST_ANY", 0) / wc, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc
        valu_rate = c.get("SQ_INSTS_VALU", 0) * VALU_CYCLES / (dur[k] * CLK * N_SIMD)
        e = {"wave_cycles_issuing": active, "wave_cycles_waiting": wait, "wave_cycles_issue_stalled": issue_stall,
             "valu_instructions_per_dispatch": c.get("SQ_INSTS_VALU", 0), "salu_instructions_per_dispatch": c.get("SQ_INSTS_SALU", 0),
             "waves_per_dispatch": c.get("SQ_WAVES", 0), "avg_dispatch_s": dur[k],
             "valu_issue_rate_of_chip": valu_rate}
        if k in SERIAL:
            e.update(bound="latency", frac=active,
                     note="one to four wavefronts per sequence run a serial Gauss-Newton chain: the fraction of the "
                          "wave's cycles in which it issues an instruction (a lone wave issues one VALU instruction "
                          "per ~4 cycles; the rest waits on LDS / L2 / dependent math). Dispatch time is the slowest "
                          "sequence of the launch, measured while the other sequence groups share the GPU")
        elif valu_rate > 0.2:
            e.update(bound="valu_issue", frac=valu_rate,
                     note="wave64 VALU instructions x 2 cycles / (dispatch time x 2.4 GHz x 1024 SIMDs), dispatch time "
                          "measured while the other sequence groups share the GPU")
        else:
            e.update(bound="occupancy_latency", frac=active,
                     note="short workgroups that wait on their first loads / barriers: fraction of wave cycles "
                          "issuing; the kernel relies on other workgroups to fill the rest, see hbm_frac in bench.py "
                          "for the bandwidth roof")
        kernels[k] = e
    return kernels


if len(sys.argv) > 3 and sys.argv[3] == "reclassify":
    # offline: <dir with pmc_sq.csv kernel_stats.csv pmc.json> <out dir> reclassify
    sq = collections.defaultdict(dict)
    for r in csv.DictReader(open(f"{src}/pmc_sq.csv")):
        sq[r["kernel"]][r["counter"]] = float(r["mean_per_dispatch"])
    old = json.load(open(f"{src}/pmc.json"))
    old["kernels"] = classify(sq, weighted_durations(f"{src}/kernel_stats.csv"))
    old["valu_instructions_per_frame"] = valu_per_frame(sq, call_counts(f"{src}/kernel_stats.csv"), old["seqs"] / old["groups"])
    if old.get("issue_activity"):
        old["issue_activity"] = issue_activity(sq, call_counts(f"{src}/kernel_stats.csv"), old["groups"], old["issue_activity"]["ms_per_step"])
    json.dump(old, open(f"{out}/pmc.json", "w"), indent=1)
    print(json.dumps({k: {"bound": v["bound"], "frac": round(v["frac"], 4)} for k, v in old["kernels"].items()}, indent=1))
    sys.exit(0)

bench = json.loads(open(f"{out}/bench_under_rocprof.json").read().strip().splitlines()[-1])
dur = weighted_durations(f"{out}/kernel_stats.csv")
sq = collect(f"{src}/sq.csv")
with open(f"{out}/pmc_sq.csv", "w") as f:
    f.write("kernel,counter,mean_per_dispatch\n")
    for k in sorted(sq):
        for c in sorted(sq[k]):
            f.write(f"{k},{c},{sq[k][c]:.1f}\n")
kernels = classify(sq, dur)
cfg = bench["config"]
meta = {"config": bench["config"]["workload"].split(":")[0], "seqs": cfg["sequences_per_gpu"], "groups": cfg["sequence_groups"]}
json.dump({**meta, "kernels": kernels,
           "valu_instructions_per_frame": valu_per_frame(sq, call_counts(f"{out}/kernel_stats.csv"), meta["seqs"] / meta["groups"]),
           "issue_activity": issue_activity(sq, call_counts(f"{out}/kernel_stats.csv"), meta["groups"], bench["ms_per_step"]),
           "note": "rocprofv3 --pmc SQ_* pass of the default bench command (own run, kernel-include-regex svo); "
                   "durations from the --kernel-trace --stats run of the same command"},
          open(f"{out}/pmc.json", "w"), indent=1)
fetch, write = collect(f"{src}/FETCH_SIZE.csv"), collect(f"{src}/WRITE_SIZE.csv")
traffic = {}
for k in set(fetch) | set(write):
    fs = fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024          # rocprofv3 reports KiB
    ws = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
    traffic[k] = {"fetch_raw": fs, "fetch_x2": 2 * fs, "write": ws, "total": 2 * fs + ws}
json.dump({**meta, "bytes_per_launch": {k: v["total"] for k, v in traffic.items()}, "detail": traffic,
           "note": "FETCH_SIZE x 2 (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section) + "
                   "WRITE_SIZE, separate --pmc passes, mean per dispatch"},
          open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps({k: {"bound": v["bound"], "frac": round(v["frac"], 4)} for k, v in kernels.items()}, indent=1))
print(json.dumps({k: round(v / 1e6, 2) for k, v in sorted(json.load(open(f"{out}/traffic.json"))["bytes_per_launch"].items())}))
