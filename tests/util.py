"""Shared helpers of the test-suite: seeded scenarios built with the CPU oracle."""
import functools

import numpy as np

import oracle_py as O
from stereo_svo_slam_amd import synth

GOLDEN = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden")


def oracle_camera(cfg):
    return O.make_camera(**{k: cfg[k] for k in synth.CAMERA_FIELDS})


@functools.lru_cache(maxsize=8)
def scenario(config="tiny", n_frames=4, seed=0, warm=1):
    """Render a sequence and run the oracle tracker over the first `warm` frames.
    Returns a dict with images (numpy), the tracker state after `warm` frames and
    everything needed to call the stage functions on frame `warm`."""
    cfg, L, R, poses, ts = synth.make_sequence(config, n_frames, seed, device="cpu")
    L = [x.numpy() for x in L]
    R = [x.numpy() for x in R]
    cam = oracle_camera(cfg)
    slam = O.Slam(cam)
    for k in range(warm):
        slam.new_image(L[k], R[k], float(ts[k]))
    k2, k3, info = slam.keypoints()
    return dict(cfg=cfg, cam=cam, L=L, R=R, poses=poses, ts=ts, slam=slam,
                kps2d=k2, kps3d=k3, info=info, warm=warm)


def flags_of(info):
    return (info["ignore_during_refinement"].astype(np.uint32) * 1 |
            info["ignore_completely"].astype(np.uint32) * 2 |
            info["ignore_temporary"].astype(np.uint32) * 4)


def real_pair():
    d = np.load(__import__("os").path.join(GOLDEN, "stereo_pair.npz"))
    return d["left"], d["right"]
