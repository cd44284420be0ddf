#!/usr/bin/env python3
"""bench.py — tracked frames/s of the stereo-SVO hot path on MI355X.

One step = StereoSlam::new_image (src/lib/stereo_slam.cpp:123-271) for every
sequence a rank owns: pyramids -> sparse image alignment -> KLT -> reprojection
GN -> SSD disparity -> depth filter (+ keyframe creation when it is due), on
frames that are already resident in HBM. Workload at N=1: BASELINE.json
configs[1] — EuRoC MH_02 class 752x480 stereo, 4-level SIA pyramid (6/2),
~110-200 patches per frame — as seeded synthetic sequences (no dataset ships),
2048 sequences per GPU in 8 sequence groups (weak scaling: the same per rank).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract of the driver). The timed region (exactly K
steps between barrier + synchronize) is repeated `--repeats` times on the running
sequences and `value` is the median repeat; all repeats are listed. `roofline`
describes the stage with the largest measured launch time (HIP events on the
stream the kernel runs on), `cpu_baseline` the CPU oracle timed on this box's
host cores (one thread, and all cores) on a bounded sample; rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

# The ctx drives every sequence group on its own HIP stream; the HIP runtime maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise. Eight
# groups need more: set before anything initialises HIP (a deployment sets it the same way,
# INTEGRATION.md section 5).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "stereo-svo-slam_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

from stereo_svo_slam_amd import multi_seq, synth
from stereo_svo_slam_amd.stereo_slam import StereoSlamBatch

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
WORKLOAD_LABEL = {"euroc": "EuRoC MH_02 class (C2)", "blender": "Blender classroom class (C1)",
                  "hd": "synthetic roofline case (C3)", "econ": "Econ Tara class, distorted (C5)", "tiny": "test size"}
KERNEL_OF_STAGE = {"sparse_align": "sia_gn_kernel", "klt": "klt_track_kernel",
                   "reproj_gn": "reproj_gn_kernel", "ssd_disparity": "ssd_disparity_kernel",
                   "filter_update": "filter_update_kernel", "images+pyramids": "pyr_fused_kernel"}
STAGES = ("images+pyramids", "compaction", "sparse_align", "klt", "reproj_gn", "ssd_disparity",
          "filter_update", "keyframe+readback")
KERNEL_STAGES = ("sparse_align", "klt", "reproj_gn", "ssd_disparity", "filter_update", "images+pyramids")
FRAMES_PER_SEQ = 24      # rendered stereo pairs per sequence; steps play them forth and back


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


def frame_index(k, n):
    """Frame shown at step k when n frames are played forth and back (a smooth path either way)."""
    if n <= 1:
        return 0
    m = k % (2 * n - 2)
    return m if m < n else 2 * n - 2 - m


def render_sequences(cfg_name, seq_ids, n_frames, device):
    """([n_seq] uint8 [n_frames,H,W] CUDA tensors left, same right); scene = id % 8, path = id.
    One launch of the fused renderer (csrc/synth_render.hip) per sequence and side."""
    cfg = dict(synth.CONFIGS[cfg_name])
    scenes = {}
    lefts, rights = [], []
    for sid in seq_ids:
        sc = scenes.setdefault(sid % 8, synth.Scene(sid % 8, device))
        poses = synth.trajectory(n_frames, sid)
        seeds = 7919 * (sid + 1) + 2 * np.arange(n_frames)
        lefts.append(synth.render_frames_gpu(sc, cfg, poses, False, 1.0, seeds))
        rights.append(synth.render_frames_gpu(sc, cfg, poses, True, 1.0, seeds + 1))
    return cfg, lefts, rights


def cpu_baseline(cfg, lefts, rights, n_play, budget_s=10.0, gpu_trajectory=None):
    """The CPU oracle (oracle/, a C restatement of the reference path) on this box's host cores,
    frames/s with the reference's formula (time inside new_image only, src/app/slam_app.cpp:186-190):
    one thread on sequence 0 — also the checker: its pose after every frame is compared with the
    HIP trajectory of that sequence — and every host core, one sequence per thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    from concurrent.futures import ThreadPoolExecutor
    cam = O.make_camera(**{k: cfg[k] for k in synth.CAMERA_FIELDS})
    nF = lefts[0].shape[0]
    ncpu = usable_cpus()
    host = {}

    def frames_of(s):
        if s not in host:
            host[s] = (lefts[s].cpu().numpy(), rights[s].cpu().numpy())
        return host[s]

    def run(s, budget, compare=None):
        L, R = frames_of(s)
        slam = O.Slam(cam)
        t_total, done, n_grad, t_sia, max_diff, kf = 0.0, 0, 0, 0.0, 0.0, 0
        for k in range(n_play):
            f = frame_index(k, nF)
            t0 = time.perf_counter()
            kf += int(slam.new_image(L[f], R[f], k / 20.0))
            dt = time.perf_counter() - t0
            if compare is not None and k < len(compare):
                max_diff = max(max_diff, float(np.max(np.abs(np.asarray(slam.pose()) - compare[k]))))
            if k > 0:                       # like the GPU leg: the first (keyframe) frame is warm-up
                t_total += dt
                done += 1
                st = slam.stats()
                n_grad += st.sia_gradient_calls
                t_sia += st.t_sia
            if t_total > budget:
                break
        return dict(frames=done, seconds=t_total, n_grad=n_grad, t_sia=t_sia, max_diff=max_diff, kf=kf)

    one = run(0, budget_s, gpu_trajectory)
    n_thr = min(ncpu, len(lefts))
    for s in range(n_thr):
        frames_of(s)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(n_thr) as ex:
        many = list(ex.map(lambda s: run(s, budget_s), range(n_thr)))
    wall = time.perf_counter() - t0
    all_fps = sum(m["frames"] for m in many) / wall if wall > 0 else None
    return {"value": one["frames"] / one["seconds"] if one["seconds"] > 0 else None, "unit": "frames/s",
            "cores": 1, "kind": "port",
            "sample": f"oracle/ (C restatement, gcc -O3, 1 thread) on sequence 0, frames 1..{one['frames']} "
                      f"of the same synthetic workload",
            "gn_ms_per_iter": 1e3 * one["t_sia"] / max(one["n_grad"], 1),
            "all_cores": {"value": all_fps, "unit": "frames/s", "cores": n_thr, "host_cpus": ncpu,
                          "sample": f"{n_thr} oracle instances, one sequence per thread, "
                                    f"{sum(m['frames'] for m in many)} frames in {wall:.1f} s wall"},
            "parity": None if gpu_trajectory is None else
            {"frames_compared": min(one["frames"] + 1, len(gpu_trajectory)), "max_abs_pose_diff": one["max_diff"],
             "tolerance": 1e-4, "keyframes_cpu": one["kf"],
             "note": "oracle pose after each frame vs the HIP trajectory of sequence 0 (m / rad)"}}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--repeats", type=int, default=3,
                    help="the timed region of exactly --steps steps is measured this many times in a row; "
                         "value = median")
    ap.add_argument("--config", default="euroc", choices=sorted(synth.CONFIGS))
    ap.add_argument("--seqs", type=int, default=2048,
                    help="sequences per GPU (default 2048 = eight groups of 256 on eight streams)")
    ap.add_argument("--fast", action="store_true",
                    help="svo_ctx_set_fast_solver(1) for the timed region: tree-ordered normal equations + LDL^T "
                         "instead of the default reference-order Gauss-Newton (bit-exact traces)")
    ap.add_argument("--copy-input", action="store_true",
                    help="SVO_MEM_DEVICE: the ctx copies every device-resident frame into its own image set "
                         "(default: SVO_MEM_DEVICE_BORROW, frames used in place)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the single-sequence and host-input legs (profiling runs)")
    ap.add_argument("--prewarm", type=float, default=1.5,
                    help="seconds of untimed load on a throw-away ctx before the warm-up steps (clock ramp)")
    ap.add_argument("--backend", default=None, choices=[None, "nccl", "gloo"],
                    help="process-group backend for --gpus > 1 (default nccl = RCCL). gloo + "
                         "--share-gpu rehearses the multi-rank path on a single-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (rehearsal only)")
    args = ap.parse_args()

    rank, local_rank, world = multi_seq.init_distributed(args.backend)
    if world != max(args.gpus, 1) and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    device = torch.device("cuda", local_rank if (world > 1 and not args.share_gpu) else 0)
    torch.cuda.set_device(device)

    K, Wm, reps = args.steps, max(args.warmup, 1), max(args.repeats, 1)
    n_steps = Wm + reps * K
    B = args.seqs
    seq_ids = multi_seq.sequence_ids(rank, world, B)
    t_setup = time.perf_counter()
    cfg, lefts, rights = render_sequences(args.config, seq_ids, FRAMES_PER_SEQ, device)
    torch.cuda.synchronize(device)
    t_setup = time.perf_counter() - t_setup
    nF = FRAMES_PER_SEQ

    # the rendered frames stay resident and unchanged for the whole run, so the ctx uses them in place
    # (SVO_MEM_DEVICE_BORROW, the reference's own level-0 alias); --copy-input ingests a copy instead
    def packs_for(slam, n, lf=None, rf=None):
        lf, rf = lf or lefts, rf or rights
        return [slam.pack_images([lf[s][frame_index(k, nF)] for s in range(slam.n)],
                                 [rf[s][frame_index(k, nF)] for s in range(slam.n)],
                                 [k / 20.0] * slam.n, borrow=not args.copy_input) for k in range(n)]

    # clocks: a fresh box starts with the GPU in a low power state. Untimed, on a throw-away ctx:
    # the same frames until --prewarm seconds have passed (only sustained load matters here)
    if args.prewarm > 0:
        warm = StereoSlamBatch(cfg, cfg["width"], cfg["height"], B, device.index)
        warm.set_fast_solver(args.fast)
        wp = packs_for(warm, 2 * nF)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < args.prewarm:      # queued like the timed steps
            for pk in wp:
                warm.submit_packed(pk)
            warm.wait()
        warm.close()
        del warm, wp

    slam = StereoSlamBatch(cfg, cfg["width"], cfg["height"], B, device.index)
    slam.enable_timing(True)
    slam.set_fast_solver(args.fast)
    packed = packs_for(slam, n_steps)
    coll_dev = device if (args.backend or "nccl") == "nccl" else None   # gloo: host tensors

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
    t0, t1 = marks[0], slam.totals()
    sec = float(np.median(seconds))
    G = max(int(t1.n_groups), 1)
    launches = max(int(t1.launches - t0.launches), 1)          # = reps * K * G: one launch of every stage each
    fps = multi_seq.throughput(B * K * world, sec)
    stage_ms = np.array(list(t1.stage_ms)) - np.array(list(t0.stage_ms))
    counters = dict(frames=t1.frames - t0.frames, keyframes=t1.keyframes - t0.keyframes,
                    n_kps=t1.keypoints - t0.keypoints,
                    n_grad=t1.gn_gradient_calls - t0.gn_gradient_calls,
                    n_cost=t1.gn_cost_calls - t0.gn_cost_calls)

    # one small exchange at the end: per-sequence summaries (id, frames, final pose)
    local = [[sid, n_steps] + [float(v) for v in slam.pose(i)] for i, sid in enumerate(seq_ids)]
    summaries = multi_seq.gather_summaries(local, world, coll_dev)

    if rank != 0:
        return
    mean_kps = counters["n_kps"] / max(counters["frames"], 1)
    ab = algorithmic_bytes(cfg, mean_kps, mean_kps)
    per_launch_ms = stage_ms / launches     # HIP events on each group's stream (svo_frame_stats.stage_ms)
    seqs_per_launch = B / G
    named = {STAGES[i]: float(per_launch_ms[i]) for i in range(8)}
    dom = max(KERNEL_STAGES, key=lambda s: named[s])           # the stage with the largest measured time
    pmc = load_profile_json("r02_pmc.json") or {}
    traffic_j = load_profile_json("r02_traffic.json") or {}
    same_cfg = lambda j: j.get("config") == args.config and j.get("seqs") == B and j.get("groups", G) == G

    def hbm_entry(stage):
        ms = named[stage]
        ach = ab[stage] * seqs_per_launch / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        e = {"kernel": KERNEL_OF_STAGE[stage], "avg_launch_ms": ms,
             "algorithmic_bytes_per_launch": ab[stage] * seqs_per_launch,
             "hbm_GBps": ach, "hbm_frac": ach / HBM_PEAK_GBS}
        b = (pmc.get("kernels") or {}).get(KERNEL_OF_STAGE[stage]) if same_cfg(pmc) else None
        if b:
            e["binding_roof"] = b
        return e

    stages = {s: hbm_entry(s) for s in KERNEL_STAGES}
    traffic = None
    if same_cfg(traffic_j):
        traffic = (traffic_j.get("bytes_per_launch") or {}).get(KERNEL_OF_STAGE[dom])
    roofline = {"kernel": KERNEL_OF_STAGE[dom], "stage": dom, "bound": "hbm",
                "achieved": stages[dom]["hbm_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": stages[dom]["hbm_frac"], "traffic": traffic,
                "algorithmic_bytes_per_launch": ab[dom] * seqs_per_launch, "avg_launch_ms": named[dom],
                "sequences_per_launch": seqs_per_launch, "stage_ms_per_launch": named,
                "binding_roof": stages[dom].get("binding_roof"),
                "stages": stages,
                "note": "dominant = the stage with the largest HIP-event launch time (events on each group's "
                        f"stream while the other {G - 1} sequence group(s) share the GPU). HBM is not the roof "
                        "that binds any of the big kernels; binding_roof (from the PMC passes under profiles/) "
                        "names the one that does",
                "frame_GBps_all_stages": sum(ab.values()) * fps / world / 1e9}
    hbm_ms = named["images+pyramids"]
    roofline["pyramids_hbm"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                "achieved": ab["images+pyramids"] * seqs_per_launch / (hbm_ms * 1e-3) / 1e9 if hbm_ms > 0 else 0.0,
                                "note": "the streaming stage: algorithmic bytes B_P of DESIGN.md section 5 / event time of the stage"}
    roofline["pyramids_hbm"]["frac"] = roofline["pyramids_hbm"]["achieved"] / HBM_PEAK_GBS
    win_d, sx_, sy_ = cfg["window_size_depth_calculator"], cfg["search_x"], cfg["search_y"]
    mfma_per_kp = win_d * ((sx_ + 1 + 15) // 16) * ((2 * sy_ + 1 + 15) // 16)
    ops = 2.0 * 16 * 16 * 64 * mfma_per_kp * mean_kps * seqs_per_launch
    roofline["ssd_mfma"] = {"bound": "mfma", "achieved": ops / (named["ssd_disparity"] * 1e-3) / 1e12,
                            "peak": 5000.0, "unit": "TOP/s (i8 dense)",
                            "frac": ops / (named["ssd_disparity"] * 1e-3) / 1e12 / 5000.0,
                            "useful_fraction_of_issued_macs": (win_d * win_d * (sx_ + 1) * (2 * sy_ + 1)) /
                                                              (16.0 * 16 * 64 * mfma_per_kp)}

    gpu_traj = np.asarray(slam.get_trajectory(0))
    slam.close()
    del packed

    single = host_input = fast_leg = None
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
        tw = time.perf_counter()
        for k in range(Wm, n1):
            one.new_images_packed(pk[k])
        tw = time.perf_counter() - tw
        b = one.totals()
        nn = n1 - Wm
        single = {"frames_per_s": nn / tw, "ms_per_frame": 1e3 * tw / nn,
                  "gn_ms_per_iter": (b.stage_ms[2] - a.stage_ms[2]) / max(b.gn_gradient_calls - a.gn_gradient_calls, 1),
                  "stage_ms_per_frame": {STAGES[i]: (b.stage_ms[i] - a.stage_ms[i]) / nn for i in range(8)}}
        one.close()
        # PCIe-inclusive leg: the same path fed from pinned host frames (SVO_MEM_HOST), 256 sequences
        Bh, nFh = min(B, 256), 8
        hl = [torch.empty((Bh, cfg["height"], cfg["width"]), dtype=torch.uint8).pin_memory() for _ in range(nFh)]
        hr = [torch.empty((Bh, cfg["height"], cfg["width"]), dtype=torch.uint8).pin_memory() for _ in range(nFh)]
        for s in range(Bh):
            for f in range(nFh):
                hl[f][s].copy_(lefts[s][f]); hr[f][s].copy_(rights[s][f])
        hs = StereoSlamBatch(cfg, cfg["width"], cfg["height"], Bh, device.index)
        hs.set_fast_solver(args.fast)
        nh = Wm + 40
        hp = [hs.pack_images([hl[frame_index(k, nFh)][s] for s in range(Bh)],
                             [hr[frame_index(k, nFh)][s] for s in range(Bh)], [k / 20.0] * Bh) for k in range(nh)]
        th = multi_seq.timed_steps(lambda k: hs.submit_packed(hp[k]), 40, Wm, 1, device, None, finish_fn=hs.wait)
        host_input = {"frames_per_s": Bh * 40 / th, "sequences": Bh,
                      "pcie_GBps": Bh * 40 / th * 2 * cfg["width"] * cfg["height"] / 1e9,
                      "note": "frames copied from pinned host memory inside the timed region; never `value`"}
        hs.close()
        del hl, hr, hp

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(cfg, lefts, rights, n_steps, gpu_trajectory=gpu_traj)

    out = {
        "metric": "tracked_frames_per_sec", "value": fps, "unit": "frames/s",
        "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": 1e3 * sec / K,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.config}: {WORKLOAD_LABEL.get(args.config, 'synthetic')} {cfg['width']}x{cfg['height']} stereo, "
                               f"{cfg['max_pyramid_levels'] - cfg['min_pyramid_level_pose_estimation']}-level SIA pyramid, "
                               f"{mean_kps:.0f} patches/frame (synthetic, seeded)",
                   "sequences_per_gpu": B, "frames_per_step": B * world,
                   "input": "device-resident frames, copied into the ctx (SVO_MEM_DEVICE)" if args.copy_input else
                            "device-resident frames used in place (SVO_MEM_DEVICE_BORROW)",
                   "solver_mode": "fast solver (tree J^T G J + LDL^T)" if args.fast else
                                  "default: reference-order Gauss-Newton (row-by-row sums + Jacobi-SVD inverse)",
                   "fast_solver_leg": fast_leg,
                   "repeats_s": seconds, "repeats_fps": [B * K * world / s for s in seconds],
                   "keyframes_in_timed_regions": counters["keyframes"],
                   "sequence_groups": G,
                   "gn_ms_per_iter": float(per_launch_ms[2] / max(counters["n_grad"] / max(counters["frames"], 1), 1e-9)),
                   "gn_gradient_calls_per_frame": counters["n_grad"] / max(counters["frames"], 1),
                   "gn_cost_calls_per_frame": counters["n_cost"] / max(counters["frames"], 1),
                   "single_sequence": single, "host_input": host_input, "setup_s": t_setup,
                   "host_cpus": os.cpu_count(), "usable_cpus": usable_cpus(),
                   "summaries_gathered": int(summaries.shape[0])},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
