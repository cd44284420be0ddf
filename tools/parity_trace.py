"""Diagnostic: frame-by-frame comparison of the HIP tracker with the oracle on one synthetic
sequence: pose after alignment / refinement / filter and the GN traces (iterations per level).
Usage: parity_trace.py config n_frames seed motion_scale [exact] [all]
Prints the frames whose GN trace or pose differs from the oracle's (every frame with `all`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stereo-svo-slam_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_py as O
import util
from stereo_svo_slam_amd import synth
from stereo_svo_slam_amd.stereo_slam import StereoSlam

config, n_frames, seed, ms = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
exact = "fast" not in sys.argv[5:]
show_all = "all" in sys.argv[5:]
cfg, L, R, poses, ts = synth.make_sequence(config, n_frames, seed, device="cuda" if torch.cuda.is_available() else "cpu",
                                           motion_scale=ms)
L = [x.cpu() for x in L]; R = [x.cpu() for x in R]
ref = O.Slam(util.oracle_camera(cfg))
gpu = StereoSlam(cfg, cfg["width"], cfg["height"])
gpu.set_exact_pinv(exact)
n_diff = 0
for k in range(n_frames):
    l, r = L[k].numpy(), R[k].numpy()
    made = ref.new_image(l, r, float(ts[k]))
    gpu.new_image(l, r, float(ts[k]))
    a, b = gpu.stats(), ref.stats()
    d_sia = np.max(np.abs(np.array(a.pose_sia) - np.array(b.pose_sia)))
    d_ref = np.max(np.abs(np.array(a.pose_refined) - np.array(b.pose_refined)))
    d_fin = np.max(np.abs(gpu.get_frame().pose - ref.pose()))
    tr = []
    same = True
    for lv in range(7, -1, -1):
        ta, tb = a.sia_trace[lv], b.sia_trace[lv]
        if ta.n_gradient or tb.n_gradient or ta.n_cost or tb.n_cost:
            eq = (ta.n_gradient, ta.n_cost, ta.n_accepted) == (tb.n_gradient, tb.n_cost, tb.n_accepted)
            same &= eq
            tr.append(f"L{lv}:{ta.n_gradient}/{tb.n_gradient},{ta.n_cost}/{tb.n_cost},"
                      f"{ta.initial_cost!r}/{tb.initial_cost!r}->{ta.final_cost!r}/{tb.final_cost!r}")
    ra, rb = a.reproj_trace, b.reproj_trace
    same &= (ra.n_gradient, ra.n_cost, ra.n_accepted) == (rb.n_gradient, rb.n_cost, rb.n_accepted)
    rp = f"rp:{ra.n_gradient}/{rb.n_gradient},{ra.n_cost}/{rb.n_cost},{ra.initial_cost!r}/{rb.initial_cost!r}->{ra.final_cost!r}/{rb.final_cost!r}"
    k2, k3, info = ref.keypoints()
    n_diff += (not same)
    if show_all or not same or d_fin > 0:
        print(f"f{k:3d} kf{made} n={a.n_keypoints}/{len(k2)} same={int(same)} dsia={d_sia:.2e} dref={d_ref:.2e} dfin={d_fin:.2e} "
              + " ".join(tr) + " " + rp, flush=True)
print(f"{config} seed {seed} x{ms} exact={exact}: {n_diff} of {n_frames - 1} tracked frames differ in GN trace")
