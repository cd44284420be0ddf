"""Wire layout of the reference's websocket backend (src/app/svo_slam_backend.cpp:18-110):
structure, key names and the robot-angle convention, on a stand-in result object (no GPU)."""
import json
from types import SimpleNamespace

import numpy as np
from scipy.spatial.transform import Rotation

from stereo_svo_slam_amd import wire
from stereo_svo_slam_amd.stereo_slam import KP_INFO_DTYPE


class _FakeSlam:
    def __init__(self):
        info = np.zeros(2, KP_INFO_DTYPE)
        info["color"][0] = (10, 20, 30)
        info["color"][1] = (40, 50, 60)
        self.kf = SimpleNamespace(pose=np.array([1, 2, 3, 0.1, -0.2, 0.3], np.float32),
                                  kps3d=np.array([[1, 2, 3], [4, 5, 6]], np.float32), info=info)
        self.traj = np.array([[0, 0, 0, 0, 0, 0], [1, 2, 3, 0.1, -0.2, 0.3]], np.float32)

    def num_keyframes(self, seq=0): return 1
    def get_keyframe(self, kid=None, seq=0): return self.kf
    def pose(self, seq=0): return self.kf.pose
    def get_trajectory(self, seq=0): return self.traj


def test_robot_angles_is_rodrigues_of_rz_rx_ry():
    pose = np.array([0, 0, 0, 0.3, -0.5, 0.7])
    rx = Rotation.from_rotvec([0.3, 0, 0]).as_matrix()
    ry = Rotation.from_rotvec([0, -0.5, 0]).as_matrix()
    rz = Rotation.from_rotvec([0, 0, 0.7]).as_matrix()
    want = Rotation.from_matrix(rz @ (rx @ ry)).as_rotvec()
    assert np.allclose(wire.robot_angles(pose), want, atol=1e-12)
    assert np.allclose(wire.robot_angles(np.zeros(6)), 0)


def test_messages_have_the_reference_layout():
    s = _FakeSlam()
    kfs = json.loads(wire.handle("ws://host:1234/keyframes", "get", s))
    assert isinstance(kfs, list) and len(kfs) == 1
    assert list(kfs[0]) == ["pose", "keypoints", "colors"]
    assert list(kfs[0]["pose"]) == ["x", "y", "z", "rx", "ry", "rz"]
    assert kfs[0]["keypoints"] == [{"x": 1.0, "y": 2.0, "z": 3.0}, {"x": 4.0, "y": 5.0, "z": 6.0}]
    assert kfs[0]["colors"] == [{"r": 10, "g": 20, "b": 30}, {"r": 40, "g": 50, "b": 60}]
    assert wire.handle("keyframes", "anything else", s) is None        # only "get" is answered
    p = json.loads(wire.handle("pose", "", s))["pose"]
    assert (p["x"], p["y"], p["z"]) == (1.0, 2.0, 3.0)
    assert np.allclose([p["rx"], p["ry"], p["rz"]], wire.robot_angles(s.kf.pose))
    t = json.loads(wire.handle("trajectory", "", s))["trajectory"]
    assert len(t) == 12 and np.allclose(t[6:], s.traj[1])               # raw angles, flat
    assert wire.handle("unknown", "get", s) is None
    assert " " not in wire.pose_message(s)                               # QJsonDocument::Compact


import pytest


@pytest.mark.gpu
def test_keyframes_message_of_a_real_tracker_carries_the_colours():
    """keyframes_message on a real StereoSlam: every keyframe keypoint arrives with its colour and
    its 3-D point (the keyframe keeps the full keypoint info, src/lib/keyframe_manager.cpp:27-29;
    the backend sends info.color per keypoint, src/app/svo_slam_backend.cpp:51-55)."""
    import oracle_py as O
    import util
    from stereo_svo_slam_amd import synth
    from stereo_svo_slam_amd.stereo_slam import StereoSlam
    cfg, L, R, poses, ts = synth.make_sequence("tiny", 14, 1, device="cpu", motion_scale=4.0)
    gpu = StereoSlam(cfg)
    ref = O.Slam(util.oracle_camera(cfg))
    for k in range(14):
        gpu.new_image(L[k].numpy(), R[k].numpy(), float(ts[k]))
        ref.new_image(L[k].numpy(), R[k].numpy(), float(ts[k]))
    kfs = json.loads(wire.keyframes_message(gpu))
    assert len(kfs) == ref.num_keyframes() >= 2
    for kid, kf in enumerate(kfs):
        k2, k3, info, pose = ref.keyframe(kid)
        assert [(c["r"], c["g"], c["b"]) for c in kf["colors"]] == [tuple(int(v) for v in c) for c in info["color"]]
        assert any(c != {"r": 0, "g": 0, "b": 0} for c in kf["colors"])
        assert np.allclose([[p["x"], p["y"], p["z"]] for p in kf["keypoints"]], k3, atol=0)
    gpu.close()
