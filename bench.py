#!/usr/bin/env python3
"""bench.py — tracked frames/s of the stereo-SVO hot path on MI355X.

One step = StereoSlam::new_image (src/lib/stereo_slam.cpp:123-271) for every
sequence a rank owns: pyramids -> sparse image alignment -> KLT -> reprojection
GN -> SSD disparity -> depth filter, plus keyframe creation whenever the
reference's rule asks for one, on frames that are already resident in HBM.
Workload at N=1: BASELINE.json configs[1] — EuRoC MH_02 class 752x480 stereo,
the reference's EuRoC.yaml settings (4-level SIA pyramid 6/2, 54x48 grid = 130
cells) — as seeded synthetic sequences (no dataset ships): closed camera loops
of 192 frames played forward, round and round, so that points leave the image
and keyframes are created at the reference's rate (every ~25 frames).
3584 sequences per GPU in 14 sequence groups of 256 (weak scaling: the same per rank; 2048 in 8
groups: about 6 % fewer frames/s at half the time per step, profiles/r03_seqs_sweep.txt).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract of the driver). The timed region (exactly K
steps between barrier + synchronize) is repeated `--repeats` times on the running
sequences and `value` is the median repeat; all repeats are listed. `roofline`
describes the stage with the largest measured launch time (HIP events on the
stream the kernel runs on), `cpu_baseline` the CPU oracle timed on this box's
host cores (one thread, and all cores) on a bounded sample; rank 0, N=1 only.
The all-cores leg runs two sequences of EVERY sequence group and its poses are
compared frame by frame with the HIP trajectories of those sequences (`parity`).
"""
import argparse
import json
import os
import sys
import time

# The ctx drives every sequence group on its own HIP stream; the HIP runtime maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise. Fourteen
# groups need more: set before anything initialises HIP (a deployment sets it the same way,
# INTEGRATION.md section 5).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "stereo-svo-slam_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

from stereo_svo_slam_amd import multi_seq, synth
from stereo_svo_slam_amd.stereo_slam import StereoSlamBatch

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
VALU_PEAK = 1024 * 2.4e9 / 2      # wave64 VALU instructions / s of the chip: 1024 SIMD-32 units, 2 cycles each (same guide)
WORKLOAD_LABEL = {"euroc": "EuRoC MH_02 class (C2)", "blender": "Blender classroom class (C1)",
                  "hd": "synthetic roofline case (C3)", "econ": "Econ Tara class, distorted (C5)", "tiny": "test size"}
KERNEL_OF_STAGE = {"sparse_align": "sia_gn_kernel", "klt": "klt_track_kernel",
                   "reproj_gn": "reproj_gn_kernel", "ssd_disparity": "ssd_disparity_kernel",
                   "filter_update": "filter_update_kernel", "images+pyramids": "pyr_fused_kernel"}
STAGES = ("images+pyramids", "compaction", "sparse_align", "klt", "reproj_gn", "ssd_disparity",
          "filter_update", "keyframe+readback")
KERNEL_STAGES = ("sparse_align", "klt", "reproj_gn", "ssd_disparity", "filter_update", "images+pyramids")
LOOP_FRAMES = 192        # rendered stereo pairs per camera loop (a closed path: frame 192 == frame 0)
MOTION_SCALE = 0.75      # synth.loop_trajectory scale: <= ~2.1 cm and <= ~1.6 deg per frame
PMC_PROFILE = "r03_pmc.json"
TRAFFIC_PROFILE = "r03_traffic.json"


def algorithmic_bytes(cfg, n, n_active):
    """ALGORITHMIC bytes of one tracked frame per stage."""
    W, H = cfg["width"], cfg["height"]
    L, lmin = cfg["max_pyramid_levels"], cfg["min_pyramid_level_pose_estimation"]
    w = cfg["window_size_depth_calculator"]
    wl = cfg["window_size_opt_flow"]
    sx, sy = cfg["search_x"], cfg["search_y"]
    b_p = 2 * W * H + sum((W >> l) * (H >> l) for l in range(1, L))
    b_p += ((W + 1) // 2) * ((H + 1) // 2) + ((W + 3) // 4) * ((H + 3) // 4)
    b_a = (L - lmin) * n_active * (64 + 36 + 20)
    b_b = n * 3 * (2 * (wl + 2) ** 2) + n * (8 + 8 + 1 + 4)
    b_r = n * (8 + 12 + 4)
    b_c = n * (w * w + (w + sx) * (w + 2 * sy)) + n * (8 + 4)
    b_d = n * (8 + 8 + 12 + 8 + 12 + 16)
    return {"images+pyramids": b_p, "sparse_align": b_a, "klt": b_b, "reproj_gn": b_r,
            "ssd_disparity": b_c, "filter_update": b_d}


def frame_index(k, n, offset=0):
    """Frame shown at step k of a closed loop of n frames entered at `offset`: always forward."""
    return (offset + k) % n if n > 0 else 0


def loop_plan(n_ctx, n_loops, loop_frames):
    """ctx sequence s plays rendered loop s % n_loops, entered at frame (s // n_loops) * spacing: the
    sequences that share a loop are `spacing` frames apart for the whole run (never the same frame
    at the same step)."""
    shares = (n_ctx + n_loops - 1) // n_loops
    spacing = max(loop_frames // shares, 1)
    return [(s % n_loops, ((s // n_loops) * spacing) % loop_frames) for s in range(n_ctx)]


def render_loops(cfg_name, loop_ids, n_frames, device, scale=MOTION_SCALE):
    """([n_loops] uint8 [n_frames,H,W] CUDA tensors left, same right); scene = id % 8, path = id.
    One launch of the fused renderer (csrc/synth_render.hip) per loop and side. `scale` is the motion
    of a LOOP_FRAMES-frame loop; a shorter loop gets a proportionally smaller path, i.e. the same
    motion per frame."""
    cfg = dict(synth.CONFIGS[cfg_name])
    scenes = {}
    lefts, rights = [], []
    for lid in loop_ids:
        sc = scenes.setdefault(lid % 8, synth.Scene(lid % 8, device))
        poses = synth.loop_trajectory(n_frames, lid, scale * n_frames / LOOP_FRAMES)
        seeds = 7919 * (lid + 1) + 2 * np.arange(n_frames)
        lefts.append(synth.render_frames_gpu(sc, cfg, poses, False, 1.0, seeds))
        rights.append(synth.render_frames_gpu(sc, cfg, poses, True, 1.0, seeds + 1))
    return cfg, lefts, rights


def group_sample(n_ctx, n_groups, per_group=2):
    """`per_group` ctx sequences of every sequence group (the ctx splits its sequences into
    contiguous groups of n/G, the first n % G one larger: svo_ctx_create), ordered so that any prefix
    covers as many groups as it can. Returns (ctx sequence ids, their group ids)."""
    firsts, first = [], 0
    for g in range(n_groups):
        count = n_ctx // n_groups + (1 if g < n_ctx % n_groups else 0)
        firsts.append((first, count))
        first += count
    seqs, groups = [], []
    for j in range(per_group):
        for g, (f0, count) in enumerate(firsts):
            if j < count:
                seqs.append(f0 + j)
                groups.append(g)
    return seqs, groups


def cpu_baseline(cfg, lefts, rights, plan, n_play, sample, sample_groups, budget_s=10.0, gpu_trajectories=None):
    """The CPU oracle (oracle/, a C restatement of the reference path) on this box's host cores,
    frames/s with the reference's formula (time inside new_image only, src/app/slam_app.cpp:186-190):
    one thread on the first sample sequence, then every usable host core, one sample sequence per
    thread. The all-cores leg is also the checker: the oracle's pose after every frame is compared
    with the HIP trajectory of the same ctx sequence (`sample` covers every sequence group)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    from concurrent.futures import ThreadPoolExecutor
    cam = O.make_camera(**{k: cfg[k] for k in synth.CAMERA_FIELDS})
    nF = lefts[0].shape[0]
    ncpu = usable_cpus()
    host = {}

    def frames_of(loop):
        if loop not in host:
            host[loop] = (lefts[loop].cpu().numpy(), rights[loop].cpu().numpy())
        return host[loop]

    def run(s, budget, compare=None):
        loop, off = plan[s]
        L, R = frames_of(loop)
        slam = O.Slam(cam)
        t_total, done, n_grad, t_sia, max_diff, kf, compared = 0.0, 0, 0, 0.0, 0.0, 0, 0
        for k in range(n_play):
            f = frame_index(k, nF, off)
            t0 = time.perf_counter()
            kf += int(slam.new_image(L[f], R[f], k / 20.0))
            dt = time.perf_counter() - t0
            if compare is not None and k < len(compare):
                max_diff = max(max_diff, float(np.max(np.abs(np.asarray(slam.pose()) - compare[k]))))
                compared += 1
            if k > 0:                       # like the GPU leg: the first (keyframe) frame is warm-up
                t_total += dt
                done += 1
                st = slam.stats()
                n_grad += st.sia_gradient_calls
                t_sia += st.t_sia
            if t_total > budget:
                break
        return dict(frames=done, seconds=t_total, n_grad=n_grad, t_sia=t_sia, max_diff=max_diff, kf=kf,
                    compared=compared)

    one = run(sample[0], budget_s)
    n_thr = min(ncpu, len(sample))
    for s in sample[:n_thr]:
        frames_of(plan[s][0])
    t0 = time.perf_counter()
    with ThreadPoolExecutor(n_thr) as ex:
        many = list(ex.map(lambda s: run(s, budget_s, None if gpu_trajectories is None else gpu_trajectories[s]),
                           sample[:n_thr]))
    wall = time.perf_counter() - t0
    all_fps = sum(m["frames"] for m in many) / wall if wall > 0 else None
    parity = None
    if gpu_trajectories is not None:
        parity = {"sequences_compared": n_thr, "groups_covered": len(set(sample_groups[:n_thr])),
                  "ctx_sequences": sample[:n_thr],
                  "frames_compared": int(sum(m["compared"] for m in many)),
                  "max_abs_pose_diff": max(m["max_diff"] for m in many), "tolerance": 0.0,
                  "keyframes_cpu": int(sum(m["kf"] for m in many)),
                  "note": "oracle pose after each frame vs the HIP trajectory of the same ctx sequence (m / rad), "
                          "two sequences of every sequence group; default solver: equal bit for bit"}
    return {"value": one["frames"] / one["seconds"] if one["seconds"] > 0 else None, "unit": "frames/s",
            "cores": 1, "kind": "port",
            "sample": f"oracle/ (C restatement, gcc -O3, 1 thread) on ctx sequence {sample[0]}, frames 1..{one['frames']} "
                      f"of the same synthetic workload",
            "gn_ms_per_iter": 1e3 * one["t_sia"] / max(one["n_grad"], 1),
            "keyframe_rate": one["kf"] / max(one["frames"] + 1, 1),
            "all_cores": {"value": all_fps, "unit": "frames/s", "cores": n_thr, "host_cpus": ncpu,
                          "sample": f"{n_thr} oracle instances, one sequence per thread, "
                                    f"{sum(m['frames'] for m in many)} frames in {wall:.1f} s wall"},
            "parity": parity}


def usable_cpus():
    """Host cores this process may use: affinity mask, cgroup quota, and the 16-per-GPU share of the box."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("SVO_BENCH_CPUS", "16"))))


def load_profile_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def run_batch(cfg, lefts, rights, plan, B, K, Wm, reps, device, world, coll_dev, fast, borrow, prewarm):
    """The timed legs of one ctx of B sequences: returns (slam, seconds per repeat, totals before / after)."""
    nF = lefts[0].shape[0]

    def packs_for(slam, n):
        return [slam.pack_images([lefts[plan[s][0]][frame_index(k, nF, plan[s][1])] for s in range(slam.n)],
                                 [rights[plan[s][0]][frame_index(k, nF, plan[s][1])] for s in range(slam.n)],
                                 [k / 20.0] * slam.n, borrow=borrow) for k in range(n)]

    # clocks: a fresh box starts with the GPU in a low power state. Untimed, on a throw-away ctx:
    # the same frames until `prewarm` seconds have passed (only sustained load matters here)
    if prewarm > 0:
        warm = StereoSlamBatch(cfg, cfg["width"], cfg["height"], B, device.index)
        warm.set_fast_solver(fast)
        wp = packs_for(warm, 24)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < prewarm:      # queued like the timed steps
            for pk in wp:
                warm.submit_packed(pk)
            warm.wait()
        warm.close()
        del warm, wp

    slam = StereoSlamBatch(cfg, cfg["width"], cfg["height"], B, device.index)
    slam.enable_timing(True)
    slam.set_fast_solver(fast)
    n_steps = Wm + reps * K
    packed = packs_for(slam, n_steps)
    # a step queues one frame set per sequence (svo_submit_images); the ctx's sequence groups
    # work through their queues independently and finish_fn (svo_wait) closes the timed region
    seconds, marks = [], [None]
    for r in range(reps):
        base = Wm + r * K if r else 0
        warm_steps = Wm if r == 0 else 0

        def step_fn(k, base=base, warm_steps=warm_steps):
            if k == warm_steps and marks[0] is None:
                marks[0] = slam.totals()
            slam.submit_packed(packed[base + k])

        seconds.append(multi_seq.timed_steps(step_fn, K, warm_steps, world, device, coll_dev, finish_fn=slam.wait))
    return slam, seconds, marks[0], slam.totals(), packs_for


def stage_table(t0, t1):
    launches = max(int(t1.launches - t0.launches), 1)          # = reps * K * G: one launch of every stage each
    stage_ms = np.array(list(t1.stage_ms)) - np.array(list(t0.stage_ms))
    per_launch_ms = stage_ms / launches     # HIP events on each group's stream (svo_frame_stats.stage_ms)
    return launches, stage_ms, {STAGES[i]: float(per_launch_ms[i]) for i in range(8)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--repeats", type=int, default=3,
                    help="the timed region of exactly --steps steps is measured this many times in a row; "
                         "value = median")
    ap.add_argument("--config", default="euroc", choices=sorted(synth.CONFIGS))
    ap.add_argument("--seqs", type=int, default=None,
                    help="sequences per GPU (default 3584 = fourteen groups of 256 on their own streams; hd: 512 = eight groups of 64)")
    ap.add_argument("--loops", type=int, default=None,
                    help="rendered camera loops per GPU (default 128; hd: 16); ctx sequence s plays loop s %% loops, "
                         "sequences that share a loop enter it at different frames")
    ap.add_argument("--loop-frames", type=int, default=None, help=f"frames per loop (default {LOOP_FRAMES}; hd: 48)")
    ap.add_argument("--motion", type=float, default=MOTION_SCALE, help="synth.loop_trajectory scale")
    ap.add_argument("--fast", action="store_true",
                    help="svo_ctx_set_fast_solver(1) for the timed region: tree-ordered normal equations + LDL^T "
                         "instead of the default reference-order Gauss-Newton (bit-exact traces)")
    ap.add_argument("--copy-input", action="store_true",
                    help="SVO_MEM_DEVICE: the ctx copies every device-resident frame into its own image set "
                         "(default: SVO_MEM_DEVICE_BORROW, frames used in place)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the fast-solver, single-sequence, host-input and C3 legs (profiling runs)")
    ap.add_argument("--c3-seqs", type=int, default=64, help="sequences of the small C3 (1920x1080) leg; 0: skip")
    ap.add_argument("--prewarm", type=float, default=1.5,
                    help="seconds of untimed load on a throw-away ctx before the warm-up steps (clock ramp)")
    ap.add_argument("--backend", default=None, choices=[None, "nccl", "gloo"],
                    help="process-group backend for --gpus > 1 (default nccl = RCCL). gloo + "
                         "--share-gpu rehearses the multi-rank path on a single-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (rehearsal only)")
    args = ap.parse_args()
    t_start = time.perf_counter()

    rank, local_rank, world = multi_seq.init_distributed(args.backend)
    if world != max(args.gpus, 1) and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    device = torch.device("cuda", local_rank if (world > 1 and not args.share_gpu) else 0)
    torch.cuda.set_device(device)

    hd = args.config == "hd"
    K, Wm, reps = args.steps, max(args.warmup, 1), max(args.repeats, 1)
    n_steps = Wm + reps * K
    B = args.seqs or (512 if hd else 3584)
    n_loops = min(B, args.loops or (16 if hd else 128))
    nF = args.loop_frames or (48 if hd else LOOP_FRAMES)
    plan = loop_plan(B, n_loops, nF)
    t_setup = time.perf_counter()
    cfg, lefts, rights = render_loops(args.config, [rank * n_loops + i for i in range(n_loops)], nF, device, args.motion)
    torch.cuda.synchronize(device)
    t_setup = time.perf_counter() - t_setup
    coll_dev = device if (args.backend or "nccl") == "nccl" else None   # gloo: host tensors

    # the rendered frames stay resident and unchanged for the whole run, so the ctx uses them in place
    # (SVO_MEM_DEVICE_BORROW, the reference's own level-0 alias); --copy-input ingests a copy instead
    slam, seconds, t0, t1, packs_for = run_batch(cfg, lefts, rights, plan, B, K, Wm, reps, device, world, coll_dev,
                                                 args.fast, not args.copy_input, args.prewarm)
    sec = float(np.median(seconds))
    G = max(int(t1.n_groups), 1)
    launches, stage_ms, named = stage_table(t0, t1)
    fps = multi_seq.throughput(B * K * world, sec)
    counters = dict(frames=t1.frames - t0.frames, keyframes=t1.keyframes - t0.keyframes,
                    n_kps=t1.keypoints - t0.keypoints,
                    n_grad=t1.gn_gradient_calls - t0.gn_gradient_calls,
                    n_cost=t1.gn_cost_calls - t0.gn_cost_calls)

    # one small exchange at the end: per-sequence summaries (id, frames, final pose)
    seq_ids = multi_seq.sequence_ids(rank, world, B)
    local = [[sid, n_steps] + [float(v) for v in slam.pose(i)] for i, sid in enumerate(seq_ids)]
    summaries = multi_seq.gather_summaries(local, world, coll_dev)

    if rank != 0:
        return
    mean_kps = counters["n_kps"] / max(counters["frames"], 1)
    kf_rate = counters["keyframes"] / max(counters["frames"], 1)
    ab = algorithmic_bytes(cfg, mean_kps, mean_kps)
    seqs_per_launch = B / G
    dom = max(KERNEL_STAGES, key=lambda s: named[s])           # the stage with the largest measured time
    pmc = load_profile_json(PMC_PROFILE) or {}
    traffic_j = load_profile_json(TRAFFIC_PROFILE) or {}
    same_cfg = lambda j: j.get("config") == args.config and j.get("seqs") == B and j.get("groups", G) == G

    def hbm_entry(stage):
        ms = named[stage]
        ach = ab[stage] * seqs_per_launch / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        e = {"kernel": KERNEL_OF_STAGE[stage], "avg_launch_ms": ms,
             "algorithmic_bytes_per_launch": ab[stage] * seqs_per_launch,
             "hbm_GBps": ach, "hbm_frac": ach / HBM_PEAK_GBS}
        b = (pmc.get("kernels") or {}).get(KERNEL_OF_STAGE[stage]) if same_cfg(pmc) else None
        if b:
            e["binding_roof"] = dict(b, source=f"profiles/{PMC_PROFILE} (an earlier rocprofv3 --pmc run of this command, "
                                                "replayed here, not measured in this run)")
        return e

    stages = {s: hbm_entry(s) for s in KERNEL_STAGES}
    traffic = None
    if same_cfg(traffic_j):
        traffic = (traffic_j.get("bytes_per_launch") or {}).get(KERNEL_OF_STAGE[dom])
    roofline = {"kernel": KERNEL_OF_STAGE[dom], "stage": dom, "bound": "hbm",
                "achieved": stages[dom]["hbm_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": stages[dom]["hbm_frac"], "traffic": traffic,
                "traffic_source": f"profiles/{TRAFFIC_PROFILE} (counter passes of an earlier run of this command, replayed)"
                if traffic is not None else None,
                "algorithmic_bytes_per_launch": ab[dom] * seqs_per_launch, "avg_launch_ms": named[dom],
                "sequences_per_launch": seqs_per_launch, "stage_ms_per_launch": named,
                "binding_roof": stages[dom].get("binding_roof"),
                "stages": stages,
                "note": "dominant = the stage with the largest HIP-event launch time (events on each group's "
                        f"stream while the other {G - 1} sequence group(s) share the GPU); achieved / frac are "
                        "ALGORITHMIC bytes over that time against the HBM peak as the contract asks. HBM does not "
                        "bind this kernel (a serial Gauss-Newton chain per sequence): binding_roof names what does, "
                        "and `valu_issue` at the top level is the chip-level figure of the whole step",
                "frame_GBps_all_stages": sum(ab.values()) * fps / world / 1e9}
    hbm_ms = named["images+pyramids"]
    roofline["pyramids_hbm"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                "achieved": ab["images+pyramids"] * seqs_per_launch / (hbm_ms * 1e-3) / 1e9 if hbm_ms > 0 else 0.0,
                                "note": "the streaming stage: algorithmic bytes B_P of DESIGN.md section 5 / event time of "
                                        "the stage (measured beside the other groups' kernels)"}
    roofline["pyramids_hbm"]["frac"] = roofline["pyramids_hbm"]["achieved"] / HBM_PEAK_GBS
    # the same kernel ALONE on the GPU (one group of 256 sequences): replayed from the committed kernel statistics
    alone_csv = os.path.join(ROOT, "profiles", "r03_kernel_stats_1group_256seq.csv")
    if args.config == "euroc" and os.path.exists(alone_csv):
        import csv
        for row in csv.DictReader(open(alone_csv)):
            if "pyr_stream_kernel" in row["Name"]:
                alone_s = float(row["AverageNs"]) * 1e-9
                roofline["pyramids_hbm"]["alone"] = {
                    "achieved": ab["images+pyramids"] * 256 / alone_s / 1e9, "frac": ab["images+pyramids"] * 256 / alone_s / 1e9 / HBM_PEAK_GBS,
                    "avg_launch_ms": alone_s * 1e3,
                    "source": "profiles/r03_kernel_stats_1group_256seq.csv (rocprofv3 --kernel-trace --stats of an earlier run with "
                              "SVO_GROUPS=1 --seqs 256, replayed; not measured in this run)"}
                break
    win_d, sx_, sy_ = cfg["window_size_depth_calculator"], cfg["search_x"], cfg["search_y"]
    mfma_per_kp = win_d * ((sx_ + 1 + 15) // 16) * ((2 * sy_ + 1 + 15) // 16)
    ops = 2.0 * 16 * 16 * 64 * mfma_per_kp * mean_kps * seqs_per_launch
    roofline["ssd_mfma"] = {"bound": "mfma", "achieved": ops / (named["ssd_disparity"] * 1e-3) / 1e12,
                            "peak": 5000.0, "unit": "TOP/s (i8 dense)",
                            "frac": ops / (named["ssd_disparity"] * 1e-3) / 1e12 / 5000.0,
                            "useful_fraction_of_issued_macs": (win_d * win_d * (sx_ + 1) * (2 * sy_ + 1)) /
                                                              (16.0 * 16 * 64 * mfma_per_kp)}
    # chip-level figure of the whole step: VALU instructions per tracked frame (counter pass under
    # profiles/) x frames/s measured here, against the chip's VALU issue rate
    valu = None
    if same_cfg(pmc) and pmc.get("valu_instructions_per_frame"):
        vpf = float(pmc["valu_instructions_per_frame"])
        valu = {"frac": vpf * fps / world / VALU_PEAK, "valu_wave_instructions_per_frame": vpf,
                "peak_wave_instructions_per_s": VALU_PEAK,
                "source": f"instructions per frame from profiles/{PMC_PROFILE} (replayed), frames/s measured in this run"}

    issue = None
    if same_cfg(pmc) and pmc.get("issue_activity"):
        issue = dict(pmc["issue_activity"], source=f"profiles/{PMC_PROFILE}: sum of SQ_ACTIVE_INST_ANY over the kernels of a step x 4 / "
                                                    "(step time x 2.4 GHz x 1024 SIMDs), an earlier counter run of this command (replayed)")
    sample, sample_groups = group_sample(B, G)
    gpu_traj = {s: np.asarray(slam.get_trajectory(s)) for s in sample}
    # fps over tracked (non-keyframe) frames: the keyframe kernels' share of the groups' event time taken out
    kf_share = float(stage_ms[7] / max(stage_ms.sum(), 1e-9))
    tracked_only = (counters["frames"] - counters["keyframes"]) / max(counters["frames"], 1) * fps / max(1.0 - kf_share, 1e-9)
    slam.close()

    single = host_input = fast_leg = c3 = None
    if not args.no_extras and world == 1 and not args.fast:
        # the same workload with svo_ctx_set_fast_solver(1): what the approximate solver would buy
        fs = StereoSlamBatch(cfg, cfg["width"], cfg["height"], B, device.index)
        fs.set_fast_solver(True)
        fp = packs_for(fs, Wm + K)
        tf = multi_seq.timed_steps(lambda k: fs.submit_packed(fp[k]), K, Wm, 1, device, None, finish_fn=fs.wait)
        fast_leg = {"frames_per_s": B * K / tf, "ms_per_step": 1e3 * tf / K,
                    "note": "svo_ctx_set_fast_solver(1): tree sums + LDL^T; pose within 1e-4 of the default "
                            "mode on smooth motion, iteration traces not the reference's; never `value`"}
        fs.close()
        del fp
    if not args.no_extras and world == 1:
        # latency leg: one sequence alone, frame by frame (svo_new_images returns when the frame is done)
        one = StereoSlamBatch(cfg, cfg["width"], cfg["height"], 1, device.index)
        one.enable_timing(True)
        one.set_fast_solver(args.fast)
        n1 = Wm + max(K, 100)
        pk = packs_for(one, n1)
        for k in range(Wm):
            one.new_images_packed(pk[k])
        torch.cuda.synchronize(device)
        a = one.totals()
        t_all, t_tracked, n_tracked = 0.0, 0.0, 0
        for k in range(Wm, n1):
            tw = time.perf_counter()
            one.new_images_packed(pk[k])
            tw = time.perf_counter() - tw
            t_all += tw
            if not one.stats(0).is_keyframe:
                t_tracked += tw
                n_tracked += 1
        b = one.totals()
        nn = n1 - Wm
        single = {"frames_per_s": nn / t_all, "ms_per_frame": 1e3 * t_all / nn,
                  "frames_per_s_tracked_frames_only": n_tracked / t_tracked if t_tracked > 0 else None,
                  "keyframes": int(b.keyframes - a.keyframes),
                  "gn_ms_per_iter": (b.stage_ms[2] - a.stage_ms[2]) / max(b.gn_gradient_calls - a.gn_gradient_calls, 1),
                  "stage_ms_per_frame": {STAGES[i]: (b.stage_ms[i] - a.stage_ms[i]) / nn for i in range(8)}}
        one.close()
        # PCIe-inclusive leg: the same path fed from pinned host frames (SVO_MEM_HOST), 256 sequences
        Bh, nFh = min(B, 256), 8
        hl = [torch.empty((Bh, cfg["height"], cfg["width"]), dtype=torch.uint8).pin_memory() for _ in range(nFh)]
        hr = [torch.empty((Bh, cfg["height"], cfg["width"]), dtype=torch.uint8).pin_memory() for _ in range(nFh)]
        for s in range(Bh):
            lp, off = plan[s]
            for f in range(nFh):
                hl[f][s].copy_(lefts[lp][(off + f) % nF]); hr[f][s].copy_(rights[lp][(off + f) % nF])
        hs = StereoSlamBatch(cfg, cfg["width"], cfg["height"], Bh, device.index)
        hs.set_fast_solver(args.fast)
        nh = Wm + 40
        fb = lambda k: k % (2 * nFh - 2) if k % (2 * nFh - 2) < nFh else 2 * nFh - 2 - k % (2 * nFh - 2)   # forth and back
        hp = [hs.pack_images([hl[fb(k)][s] for s in range(Bh)], [hr[fb(k)][s] for s in range(Bh)], [k / 20.0] * Bh)
              for k in range(nh)]
        th = multi_seq.timed_steps(lambda k: hs.submit_packed(hp[k]), 40, Wm, 1, device, None, finish_fn=hs.wait)
        host_input = {"frames_per_s": Bh * 40 / th, "sequences": Bh,
                      "pcie_GBps": Bh * 40 / th * 2 * cfg["width"] * cfg["height"] / 1e9,
                      "note": "frames copied from pinned host memory inside the timed region; never `value`"}
        hs.close()
        del hl, hr, hp

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(cfg, lefts, rights, plan, n_steps, sample, sample_groups, gpu_trajectories=gpu_traj)

    if not args.no_extras and world == 1 and not hd and args.c3_seqs > 0:
        # a small run of BASELINE config 3 (1920x1080, ~2000 patches, 5-level SIA pyramid) so that the
        # driver's line carries it; the full-size C3 run is `bench.py --config hd` (profiles/)
        del lefts, rights
        torch.cuda.empty_cache()
        c3 = c3_leg(args.c3_seqs, device, Wm)

    out = {
        "metric": "tracked_frames_per_sec", "value": fps, "unit": "frames/s",
        "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": 1e3 * sec / K,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        # scalars of the nested legs, repeated at the top level
        "patches_per_frame": mean_kps, "keyframe_rate": kf_rate,
        "fps_all_frames": fps, "fps_tracked_frames_only": tracked_only,
        "gn_ms_per_iter": float(named["sparse_align"] / max(counters["n_grad"] / max(counters["frames"], 1), 1e-9)),
        "single_sequence_fps": single and single["frames_per_s"],
        "host_input_fps": host_input and host_input["frames_per_s"],
        "fast_solver_fps": fast_leg and fast_leg["frames_per_s"],
        "cpu_one_core_fps": cpu and cpu["value"], "cpu_all_cores_fps": cpu and cpu["all_cores"]["value"],
        "speedup_vs_one_core": (fps / cpu["value"]) if cpu and cpu["value"] else None,
        "single_sequence_speedup_vs_one_core": (single["frames_per_s"] / cpu["value"]) if cpu and cpu["value"] and single else None,
        "parity_max_abs_pose_diff": cpu and cpu["parity"] and cpu["parity"]["max_abs_pose_diff"],
        "parity_sequences_compared": cpu and cpu["parity"] and cpu["parity"]["sequences_compared"],
        "parity_groups_covered": cpu and cpu["parity"] and cpu["parity"]["groups_covered"],
        "valu_issue_frac_of_step": valu and valu["frac"],
        "issue_active_frac_of_step": issue and issue["per_simd_cycle"],
        "pyramids_hbm_frac_alone_replayed": roofline["pyramids_hbm"].get("alone", {}).get("frac"),
        "c3_fps": c3 and c3["frames_per_s"], "c3_pyramids_hbm_frac": c3 and c3["pyramids_hbm_frac"],
        "setup_s": t_setup, "wall_s": time.perf_counter() - t_start,
        "config": {"workload": f"{args.config}: {WORKLOAD_LABEL.get(args.config, 'synthetic')} {cfg['width']}x{cfg['height']} stereo, "
                               f"{cfg['max_pyramid_levels'] - cfg['min_pyramid_level_pose_estimation']}-level SIA pyramid, "
                               f"{mean_kps:.0f} patches/frame, a keyframe every {1 / max(kf_rate, 1e-9):.0f} frames "
                               f"(synthetic, seeded: {n_loops} closed camera loops of {nF} frames played forward, "
                               f"motion scale {args.motion})",
                   "sequences_per_gpu": B, "frames_per_step": B * world,
                   "rendered_loops": n_loops, "frames_per_loop": nF,
                   "sequences_per_loop": (B + n_loops - 1) // n_loops,
                   "input": "device-resident frames, copied into the ctx (SVO_MEM_DEVICE)" if args.copy_input else
                            "device-resident frames used in place (SVO_MEM_DEVICE_BORROW)",
                   "solver_mode": "fast solver (tree J^T G J + LDL^T)" if args.fast else
                                  "default: reference-order Gauss-Newton (row-by-row sums + Jacobi-SVD inverse)",
                   "fast_solver_leg": fast_leg,
                   "repeats_s": seconds, "repeats_fps": [B * K * world / s for s in seconds],
                   "keyframes_in_timed_regions": counters["keyframes"],
                   "keyframe_stage_share_of_event_time": kf_share,
                   "fps_tracked_frames_only_note": "non-keyframe frames / (time x (1 - share of the keyframe stage in the "
                                                   "groups' HIP-event time)): an estimate; the single-sequence leg measures it directly",
                   "sequence_groups": G,
                   "image_sets_allocated": int(t1.image_sets), "keyframes_created": int(t1.keyframes),
                   "image_sets_note": "image sets (pyramids of one frame) the ctx allocated in the whole run, beside the keyframes it "
                                      "created: a keyframe's images are released once no tracked keypoint comes from it",
                   "gn_gradient_calls_per_frame": counters["n_grad"] / max(counters["frames"], 1),
                   "gn_cost_calls_per_frame": counters["n_cost"] / max(counters["frames"], 1),
                   "single_sequence": single, "host_input": host_input, "c3": c3,
                   "host_cpus": os.cpu_count(), "usable_cpus": usable_cpus(),
                   "summaries_gathered": int(summaries.shape[0])},
        "roofline": roofline, "valu_issue": valu, "issue_activity": issue, "cpu_baseline": cpu,
    }
    print(json.dumps(out))


def c3_leg(n_seq, device, warm):
    """BASELINE config 3 at a small batch inside the default run: frames/s, patches per frame and the
    pyramid stage's share of the HBM peak."""
    n_loops, nF, K = min(n_seq, 8), 24, 12
    plan = loop_plan(n_seq, n_loops, nF)
    cfg, lefts, rights = render_loops("hd", [5000 + i for i in range(n_loops)], nF, device, MOTION_SCALE)
    torch.cuda.synchronize(device)
    slam, seconds, t0, t1, _ = run_batch(cfg, lefts, rights, plan, n_seq, K, warm, 1, device, 1, None, False, True, 0.0)
    launches, stage_ms, named = stage_table(t0, t1)
    frames = t1.frames - t0.frames
    G = max(int(t1.n_groups), 1)
    kps = (t1.keypoints - t0.keypoints) / max(frames, 1)
    ab = algorithmic_bytes(cfg, kps, kps)
    slam.close()
    pyr = ab["images+pyramids"] * (n_seq / G) / (named["images+pyramids"] * 1e-3) / 1e9 if named["images+pyramids"] > 0 else 0.0
    return {"frames_per_s": n_seq * K / seconds[0], "sequences": n_seq, "sequence_groups": G, "steps": K,
            "patches_per_frame": kps, "keyframe_rate": (t1.keyframes - t0.keyframes) / max(frames, 1),
            "stage_ms_per_launch": named, "pyramids_hbm_GBps": pyr, "pyramids_hbm_frac": pyr / HBM_PEAK_GBS,
            "sparse_align_share_of_event_time": float(stage_ms[2] / max(stage_ms.sum(), 1e-9)),
            "note": "hd: 1920x1080, 43x24 grid, 5-level SIA pyramid; small batch inside the default run "
                    "(the full C3 run: bench.py --config hd, profiles/)"}


if __name__ == "__main__":
    main()
