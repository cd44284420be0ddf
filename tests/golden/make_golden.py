"""Generates tests/golden/golden_real_pair.npz with the CPU oracle.

Inputs : tests/golden/stereo_pair.npz — the real Econ-Tara 752x480 stereo pair the
         reference's own tests hold (src/test/left.png, right.png, testimage0.png; data
         files, converted to uint8 arrays).
Outputs: per-stage results of the oracle on that pair (pyramid checksums, LK pyramid
         checksums, SSD disparities, KLT tracks, FAST/edgelet grid keypoints, first-frame
         keyframe) — vectors for the oracle-regression test and the GPU parity tests.
Run    : python tests/golden/make_golden.py   (from the repository root)
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "stereo-svo-slam_amd")]
import oracle_py as O                                   # noqa: E402
from stereo_svo_slam_amd import synth                   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def main():
    d = np.load(os.path.join(HERE, "stereo_pair.npz"))
    left, right = d["left"], d["right"]
    out = {}
    pyr = O.build_pyramid(left, 6)
    out["pyr_crc"] = np.array([crc(p) for p in pyr], np.uint32)
    lk = O.build_lk_pyramid(left, 31)
    out["lk_crc"] = np.array([crc(p) for p in lk], np.uint32)
    out["scharr_crc"] = np.array([crc(O.scharr(p)) for p in lk], np.uint32)
    rng = np.random.RandomState(2024)
    kps = np.stack([rng.uniform(0, 752, 200), rng.uniform(0, 480, 200)], 1).astype(np.float32)
    out["kps"] = kps
    out["disp_31"] = O.ssd_disparity(left, right, kps, 31, 60, 6, 1)
    out["disp_35"] = O.ssd_disparity(left, right, kps, 35, 60, 6, 1)
    init = (kps + np.float32([10, 0])).astype(np.float32)
    lkr = O.build_lk_pyramid(right, 31)
    pts, st, err = O.klt_track(lk, lkr, kps, init, 31)
    out["klt_pts"], out["klt_status"], out["klt_err"] = pts, st, err
    out["fast_crc"] = crc(O.fast_score_nms(left, 6))
    out["sobel_crc"] = crc(O.sobel_x_u8(left))
    dk, ds, dt = O.detect_keypoints(left, 40, 50, 0)
    out["det_kps"], out["det_score"], out["det_type"] = dk, ds, dt
    cfg = dict(synth.CONFIGS["econ"])
    slam = O.Slam(O.make_camera(**{k: cfg[k] for k in synth.CAMERA_FIELDS}))
    slam.new_image(left, right, 0.0)
    k2, k3, info = slam.keypoints()
    out["kf0_kps2d"], out["kf0_kps3d"] = k2, k3
    out["kf0_type"], out["kf0_level"] = info["type"], info["level"]
    np.savez_compressed(os.path.join(HERE, "golden_real_pair.npz"), **out)
    print("wrote golden_real_pair.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
