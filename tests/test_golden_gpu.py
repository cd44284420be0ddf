"""HIP path against the COMMITTED golden vectors (tests/golden/*.npz, written by
tests/golden/make_golden.py with the CPU oracle on the reference's own stereo pair and on a
stored 5-frame sequence). No oracle is imported here: a change that moved the oracle and the
kernels together would pass the live-oracle tests and fail these."""
import os
import sys
import zlib

import numpy as np
import pytest
import torch

from stereo_svo_slam_amd import hip_lib, synth
from stereo_svo_slam_amd.stereo_slam import StereoSlam

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def flags_of(info):
    return (info["ignore_during_refinement"].astype(np.uint32) * 1 | info["ignore_completely"].astype(np.uint32) * 2 |
            info["ignore_temporary"].astype(np.uint32) * 4)


@pytest.fixture(scope="module")
def H():
    assert "oracle_py" not in sys.modules or True   # (other test modules of the session may have loaded it)
    h = hip_lib.Handle(0, max_keypoints=4096)
    yield h
    h.close()


def test_stage_entries_against_golden_real_pair(H):
    g = np.load(os.path.join(GOLDEN, "golden_real_pair.npz"))
    pair = np.load(os.path.join(GOLDEN, "stereo_pair.npz"))
    left, right = pair["left"], pair["right"]
    # integer stages: bit exact
    pyr = H.build_pyramid(dev(left), 6)
    assert np.array_equal(np.array([crc(p.cpu().numpy()) for p in pyr], np.uint32), g["pyr_crc"])
    lk = H.build_lk_pyramid(dev(left), 31)
    assert np.array_equal(np.array([crc(p.cpu().numpy()) for p in lk], np.uint32), g["lk_crc"])
    for win, key in ((31, "disp_31"), (35, "disp_35")):
        d = H.ssd_disparity(dev(left), dev(right), dev(g["kps"]), win, 60, 6, 1)
        assert np.array_equal(d.cpu().numpy(), g[key])
    cur = dev((g["kps"] + np.float32([10, 0])).astype(np.float32))
    _, st, err = H.klt_track(lk, H.build_lk_pyramid(dev(right), 31), dev(g["kps"]), cur, 31)
    assert np.array_equal(st.cpu().numpy(), g["klt_status"])
    assert np.array_equal(cur.cpu().numpy(), g["klt_pts"]) and np.array_equal(err.cpu().numpy(), g["klt_err"])
    # first-frame keyframe of the tracker (FAST / edgelet grid, merge, SSD depth)
    cfg = dict(synth.CONFIGS["econ"])
    slam = StereoSlam(cfg)
    slam.new_image(left, right, 0.0)
    f = slam.get_frame()
    assert np.array_equal(f.kps2d, g["kf0_kps2d"]) and np.array_equal(f.kps3d, g["kf0_kps3d"])
    assert np.array_equal(f.info["type"], g["kf0_type"]) and np.array_equal(f.info["level"], g["kf0_level"])
    slam.close()
    # sparse image alignment left -> right, reference-order mode: the golden trace and pose
    cam = hip_lib.CameraSettings.from_dict(cfg)
    nl = cfg["max_pyramid_levels"]
    info = g["kf0_info"]
    k2, k3, fl = dev(g["kf0_kps2d"]), dev(g["kf0_kps3d"]), dev(flags_of(info))
    H.set_exact_pinv(True)
    pose, cost, trace, _ = H.sparse_align(H.build_pyramid(dev(left), nl), H.build_pyramid(dev(right), nl), k2, k3, fl,
                                          cam, dev(np.zeros(6, np.float32)))
    tr = hip_lib.trace_to_numpy(trace)
    got = np.array([[t["n_gradient"], t["n_cost"], t["n_accepted"], t["exit_small"]] for t in tr[:nl]], np.int32)
    assert np.array_equal(got, g["sia_trace"]), (got, g["sia_trace"])
    assert np.array_equal(pose.cpu().numpy(), g["sia_pose"]), (pose, g["sia_pose"])
    assert float(cost.cpu()) == float(g["sia_cost"])
    # merge + reprojection GN
    m2, mfl = dev(g["rp_proj0"].copy()), fl.clone()
    rp, rcost, rtrace = H.reproj_gn(m2, k3, mfl, cam, dev(g["rp_start"]), dev(g["rp_tracked"]), dev(g["rp_err"]))
    assert np.array_equal(m2.cpu().numpy(), g["rp_merged"]) and np.array_equal(mfl.cpu().numpy(), g["rp_flags"])
    t = hip_lib.trace_to_numpy(rtrace)[0]
    assert [int(t["n_gradient"]), int(t["n_cost"]), int(t["n_accepted"]), int(t["exit_small"])] == g["rp_trace"].tolist()
    assert np.array_equal(rp.cpu().numpy(), g["rp_pose"]) and float(rcost.cpu()) == float(g["rp_cost"])
    # fast solver mode: same minimum within the stated tolerance
    H.set_fast_solver(True)
    pose_d, _, _, _ = H.sparse_align(H.build_pyramid(dev(left), nl), H.build_pyramid(dev(right), nl), k2, k3, fl,
                                     cam, dev(np.zeros(6, np.float32)))
    H.set_fast_solver(False)
    assert np.max(np.abs(pose_d.cpu().numpy() - g["sia_pose"])) < 5e-4


def test_tracker_against_golden_sequence():
    s = np.load(os.path.join(GOLDEN, "golden_sequence.npz"))
    cfg = dict(synth.CONFIGS["tiny"])
    slam = StereoSlam(cfg)                               # default mode = reference order
    fields = ("level", "type", "keyframe_id", "keypoint_index", "ignore_during_refinement", "ignore_completely",
              "ignore_temporary", "outlier_count", "inlier_count", "score", "color")
    for k in range(len(s["ts"])):
        slam.new_image(s["left"][k], s["right"][k], float(s["ts"][k]))
        st = slam.stats()
        f = slam.get_frame()
        info = s[f"f{k}_info"]
        assert st.is_keyframe == int(s[f"f{k}_kf"])
        assert len(f.kps2d) == len(info)
        for fld in fields:                                  # feature index lists: bit exact
            assert np.array_equal(f.info[fld], info[fld]), (k, fld)
        trace = np.array([[t.n_gradient, t.n_cost, t.n_accepted, t.exit_small] for t in st.sia_trace] +
                         [[st.reproj_trace.n_gradient, st.reproj_trace.n_cost, st.reproj_trace.n_accepted,
                           st.reproj_trace.exit_small]], np.int32)
        assert np.array_equal(trace, s[f"f{k}_trace"]), (k, trace.tolist(), s[f"f{k}_trace"].tolist())
        # reference-order mode reproduces the oracle's floats: pose and points equal
        assert np.array_equal(f.pose, s[f"f{k}_pose"]), (k, f.pose, s[f"f{k}_pose"])
        assert np.array_equal(f.kps2d, s[f"f{k}_kps2d"]) and np.array_equal(f.kps3d, s[f"f{k}_kps3d"]), k
    slam.close()
