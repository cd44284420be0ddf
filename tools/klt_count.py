"""Diagnostic: one klt_track launch on a fixed input (the real stereo pair, 4096 points, initial flow
off by ~5 px) — run under rocprofv3 --pmc to count instructions per keypoint."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "stereo-svo-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from stereo_svo_slam_amd import hip_lib, synth
win = int(sys.argv[1]) if len(sys.argv) > 1 else 31
cfg, L, R, poses, ts = synth.make_sequence_gpu("euroc", 2, 3, motion_scale=3.0)
H = hip_lib.Handle(0, 8192)
gp, gc = H.build_lk_pyramid(L[0].contiguous(), win), H.build_lk_pyramid(L[1].contiguous(), win)
rng = np.random.RandomState(1)
n = 4096
pts = np.stack([rng.uniform(40, 712, n), rng.uniform(40, 440, n)], 1).astype(np.float32)
init = pts + rng.uniform(-3, 3, pts.shape).astype(np.float32)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for rep in range(3):
    cp = d(init.copy())
    _, st, err = H.klt_track(gp, gc, d(pts), cp, win)
    torch.cuda.synchronize()
print("tracked", int(st.sum().item()), "of", n)
