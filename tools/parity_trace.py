"""Diagnostic: frame-by-frame comparison of the HIP tracker with the oracle on one synthetic
sequence: pose after alignment / refinement / filter and the GN traces (iterations per level).
Usage: parity_trace.py config n_frames seed motion_scale [exact]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stereo-svo-slam_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_py as O
import util
from stereo_svo_slam_amd import synth
from stereo_svo_slam_amd.stereo_slam import StereoSlam

config, n_frames, seed, ms = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
exact = len(sys.argv) > 5 and sys.argv[5] == "exact"
cfg, L, R, poses, ts = synth.make_sequence(config, n_frames, seed, device="cpu", motion_scale=ms)
ref = O.Slam(util.oracle_camera(cfg))
gpu = StereoSlam(cfg, cfg["width"], cfg["height"])
gpu.set_exact_pinv(exact)
for k in range(n_frames):
    l, r = L[k].numpy(), R[k].numpy()
    made = ref.new_image(l, r, float(ts[k]))
    gpu.new_image(l, r, float(ts[k]))
    a, b = gpu.stats(), ref.stats()
    d_sia = np.max(np.abs(np.array(a.pose_sia) - np.array(b.pose_sia)))
    d_ref = np.max(np.abs(np.array(a.pose_refined) - np.array(b.pose_refined)))
    d_fin = np.max(np.abs(gpu.get_frame().pose - ref.pose()))
    tr = []
    for lv in range(8):
        ta, tb = a.sia_trace[lv], b.sia_trace[lv]
        if ta.n_gradient or tb.n_gradient:
            tr.append(f"L{lv}:{ta.n_gradient}/{tb.n_gradient},{ta.n_cost}/{tb.n_cost},{ta.final_cost:.1f}/{tb.final_cost:.1f}")
    rp = f"rp:{a.reproj_trace.n_gradient}/{b.reproj_trace.n_gradient},{a.reproj_trace.n_cost}/{b.reproj_trace.n_cost}"
    k2, k3, info = ref.keypoints()
    print(f"f{k:3d} kf{made} n={a.n_keypoints}/{len(k2)} dsia={d_sia:.2e} dref={d_ref:.2e} dfin={d_fin:.2e} " + " ".join(tr) + " " + rp, flush=True)
