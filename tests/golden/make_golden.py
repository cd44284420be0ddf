"""Generates tests/golden/golden_real_pair.npz with the CPU oracle.

Inputs : tests/golden/stereo_pair.npz — the real Econ-Tara 752x480 stereo pair the
         reference's own tests hold (src/test/left.png, right.png, testimage0.png; data
         files, converted to uint8 arrays).
Outputs: per-stage results of the oracle on that pair (pyramid checksums, LK pyramid
         checksums, SSD disparities, KLT tracks, FAST/edgelet grid keypoints, first-frame
         keyframe, sparse alignment left -> right, merge + reprojection GN) and, in
         golden_sequence.npz, a 5-frame synthetic sequence (images included) with the oracle
         tracker's pose, keypoints and GN traces per frame — vectors for the oracle-regression
         test (tests/test_oracle_cpu.py) and the GPU tests that compare the HIP path with these
         files alone (tests/test_golden_gpu.py).
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
    out["kf0_info"] = info
    # sparse image alignment of the pair itself: the right image is the left camera moved along x,
    # so aligning left -> right from a zero guess must find that translation (levels 4..2)
    cam = O.make_camera(**{k: cfg[k] for k in synth.CAMERA_FIELDS})
    nl = cfg["max_pyramid_levels"]
    fl = (info["ignore_during_refinement"].astype(np.uint32) | info["ignore_completely"].astype(np.uint32) * 2 |
          info["ignore_temporary"].astype(np.uint32) * 4)
    pose, cost, tr = O.sparse_align(O.build_pyramid(left, nl), O.build_pyramid(right, nl), k2, k3, fl, cam,
                                    np.zeros(6, np.float32))
    out["sia_pose"], out["sia_cost"] = pose, np.float32(cost)
    out["sia_trace"] = np.array([[t["n_gradient"], t["n_cost"], t["n_accepted"], t["exit_small"]] for t in tr], np.int32)
    out["sia_costs"] = np.array([[t["initial_cost"], t["final_cost"]] for t in tr], np.float32)
    # merge + reprojection GN on seeded observations around that pose
    rng = np.random.RandomState(77)
    proj = O.project_keypoints(pose, k3, cam)
    tracked = (proj + rng.normal(0, 0.3, proj.shape)).astype(np.float32)
    tracked[::17] += 15.0
    err = rng.uniform(0, 10, len(k3)).astype(np.float32)
    err[::13] = 30.0
    start = (pose + np.float32([0.01, -0.005, 0.008, 0.002, -0.001, 0.0015])).astype(np.float32)
    proj0 = O.project_keypoints(start, k3, cam)
    m2, mfl = O.refine_merge(proj0, fl, tracked, err)
    rpose, rcost, rtr = O.reproj_gn(m2, k3, mfl, cam, start)
    out["rp_tracked"], out["rp_err"], out["rp_start"], out["rp_proj0"] = tracked, err, start, proj0
    out["rp_merged"], out["rp_flags"], out["rp_pose"], out["rp_cost"] = m2, mfl, rpose, np.float32(rcost)
    out["rp_trace"] = np.array([rtr["n_gradient"], rtr["n_cost"], rtr["n_accepted"], rtr["exit_small"]], np.int32)
    np.savez_compressed(os.path.join(HERE, "golden_real_pair.npz"), **out)
    print("wrote golden_real_pair.npz:", {k: v.shape for k, v in out.items()})

    # a short tracked sequence (images stored: the renderer is not part of what is pinned)
    cfg, L, R, poses, ts = synth.make_sequence("tiny", 5, 0, device="cpu", motion_scale=2.0)
    slam = O.Slam(O.make_camera(**{k: cfg[k] for k in synth.CAMERA_FIELDS}))
    seq = {"left": np.stack([x.numpy() for x in L]), "right": np.stack([x.numpy() for x in R]), "ts": ts}
    for k in range(len(L)):
        kf = slam.new_image(seq["left"][k], seq["right"][k], float(ts[k]))
        k2, k3, info = slam.keypoints()
        st = slam.stats()
        seq[f"f{k}_pose"], seq[f"f{k}_kps2d"], seq[f"f{k}_kps3d"], seq[f"f{k}_info"] = slam.pose(), k2, k3, info
        seq[f"f{k}_kf"] = np.int32(kf)
        seq[f"f{k}_trace"] = np.array(
            [[t.n_gradient, t.n_cost, t.n_accepted, t.exit_small] for t in st.sia_trace] +
            [[st.reproj_trace.n_gradient, st.reproj_trace.n_cost, st.reproj_trace.n_accepted,
              st.reproj_trace.exit_small]], np.int32)
    np.savez_compressed(os.path.join(HERE, "golden_sequence.npz"), **seq)
    print("wrote golden_sequence.npz", os.path.getsize(os.path.join(HERE, "golden_sequence.npz")), "bytes")


if __name__ == "__main__":
    main()
