"""Replay harness: YAML reader, trajectory CSV, FPS formula, error report."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from stereo_svo_slam_amd import replay, synth

YAML = """%YAML:1.0
# keys of src/app/image_input.cpp:16-36 (values: a EuRoC-like rig)
Camera1.fx: 435.2046959714599
Camera1.fy: 435.2046959714599
Camera1.cx: 367.4517211914062 # principal point
Camera1.cy: 252.2008514404297
Camera.baseline: 47.90639384423901
Camera1.k1: 0.0
Camera1.k2: -0.25
Camera1.k3: 0.0
Camera1.p1: 1e-3
Camera1.p2: 0.0
Camera.width: 752
Camera.height: 480
Camera.grid_width: 54
Camera.grid_height: 48
Camera.search_x: 60
Camera.search_y: 6
Camera.window_size_pose_estimator: 4
Camera.window_size_opt_flow: 31
Camera.window_size_depth_calculator: 31
Camera.max_pyramid_levels: 6
Camera.min_pyramid_level_pose_estimation: 2
LEFT.K: !!opencv-matrix
   rows: 3
   cols: 3
   dt: d
   data: [458.654, 0.0, 367.215, 0.0, 457.296, 248.375, 0.0, 0.0, 1.0]
"""


def test_read_settings(tmp_path):
    p = tmp_path / "cam.yaml"
    p.write_text(YAML)
    s = replay.read_settings(str(p))
    assert s["fx"] == pytest.approx(435.2046959714599) and s["baseline"] == pytest.approx(47.90639384423901)
    assert s["grid_width"] == 54 and s["grid_height"] == 48 and s["max_pyramid_levels"] == 6
    assert s["k2"] == -0.25 and s["p1"] == 1e-3 and s["width"] == 752 and s["height"] == 480
    for f in synth.CAMERA_FIELDS:
        assert f in s


def test_csv_angles_is_ry_rx_rz():
    rng = np.random.RandomState(0)
    for _ in range(20):
        pose = np.concatenate([rng.normal(0, 1, 3), rng.normal(0, 0.4, 3)])
        R = (Rotation.from_rotvec([0, pose[4], 0]) * Rotation.from_rotvec([pose[3], 0, 0]) *
             Rotation.from_rotvec([0, 0, pose[5]]))
        assert np.allclose(replay.csv_angles(pose), R.as_rotvec(), atol=1e-9)
    assert np.allclose(replay.csv_angles(np.zeros(6)), 0)


def test_csv_and_fps_formula(tmp_path):
    traj = np.array([[0, 0, 0, 0, 0, 0], [0.1, 0.2, 0.3, 0.01, 0.02, 0.03]], np.float32)
    out = tmp_path / "t.csv"
    replay.write_trajectory_csv(str(out), [0.02, 0.05], traj)
    rows = np.loadtxt(str(out), delimiter=",")
    assert rows.shape == (2, 7) and np.allclose(rows[1, 1:4], [0.1, 0.2, 0.3], atol=1e-6)
    assert np.allclose(rows[1, 4:], replay.csv_angles(traj[1]), atol=1e-6)
    # test/extract_fps.py: n / (t_last - t_first)
    assert replay.fps_from_csv_rows(rows) == pytest.approx(2 / 0.03)
    rep = replay.error_report(rows, rows + np.array([0, 0.01, 0, 0, np.pi / 180, 0, 0]))
    assert rep["max"][0] == pytest.approx(0.01) and rep["max"][3] == pytest.approx(1.0)


@pytest.mark.gpu
def test_replay_synthetic_sequence(tmp_path):
    out = tmp_path / "traj.csv"
    replay.main(["--synthetic", "tiny", "--frames", "10", "-t", str(out)])
    rows = np.loadtxt(str(out), delimiter=",")
    assert rows.shape == (10, 7)
    assert np.all(np.diff(rows[:, 0]) > 0)                   # cumulative algorithm time
    gt = synth.trajectory(10, 0)
    assert np.max(np.abs(rows[:, 1:4] - gt[:, :3])) < 0.06   # same envelope as the oracle test
