"""Diagnostic: one ssd_disparity launch on a fixed input (synthetic euroc pair, 4096 keypoints) — run
under rocprofv3 --pmc to count instructions per keypoint workgroup."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "stereo-svo-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from stereo_svo_slam_amd import hip_lib, synth
cfg, L, R, poses, ts = synth.make_sequence_gpu("euroc", 1, 3)
H = hip_lib.Handle(0, 8192)
rng = np.random.RandomState(1)
n = 4096
pts = np.stack([rng.uniform(40, 712, n), rng.uniform(40, 440, n)], 1).astype(np.float32)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for rep in range(3):
    disp = H.ssd_disparity(L[0].contiguous(), R[0].contiguous(), d(pts), 31, 60, 6, 1)
    torch.cuda.synchronize()
print("mean disparity", float(disp.mean().item()))
