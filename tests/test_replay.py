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


RECT_YAML = """%YAML:1.0
Camera1.fx: 200.0
Camera1.fy: 200.0
Camera1.cx: 160.0
Camera1.cy: 120.0
Camera.baseline: 20.0
Camera.grid_width: 40
Camera.grid_height: 40
Camera.search_x: 30
Camera.search_y: 4
Camera.window_size_pose_estimator: 4
Camera.window_size_opt_flow: 21
Camera.window_size_depth_calculator: 21
Camera.max_pyramid_levels: 4
Camera.min_pyramid_level_pose_estimation: 1
LEFT.height: 240
LEFT.width: 320
RIGHT.height: 240
RIGHT.width: 320
{mats}
"""


def _mat(key, rows, cols, data):
    return (f"{key}: !!opencv-matrix\n   rows: {rows}\n   cols: {cols}\n   dt: d\n   data: [" +
            ", ".join(repr(float(v)) for v in data) + "]\n")


def _write_euroc(tmp_path, k_shift=0.0):
    """A three-frame EuRoC-layout dataset (mav0/cam0, cam1 + data.csv) of the synthetic tiny scene and
    a settings file whose rectification is the identity (optionally: principal point moved by k_shift)."""
    from PIL import Image
    cfg, L, R, poses, ts = synth.make_sequence("tiny", 3, 0, device="cpu")
    mav = tmp_path / "mav0"
    for cam in ("cam0", "cam1"):
        (mav / cam / "data").mkdir(parents=True)
    lines = ["#timestamp [ns],filename"]
    for k in range(3):
        stamp = 1403636579763555584 + k * 50000000
        # cam0 is the physically left camera = the library's `right`; cam1 the library's `left`
        Image.fromarray(R[k].numpy()).save(str(mav / "cam0" / "data" / f"{stamp}.png"))
        Image.fromarray(L[k].numpy()).save(str(mav / "cam1" / "data" / f"{stamp}.png"))
        lines.append(f"{stamp},{stamp}.png")
    (mav / "cam0" / "data.csv").write_text("\r\n".join(lines) + "\r\n")
    K = [200.0, 0, 160.0 + k_shift, 0, 200.0, 120.0, 0, 0, 1]
    P = [200.0, 0, 160.0, 0, 0, 200.0, 120.0, 0, 0, 0, 1, 0]
    eye = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    mats = ""
    for side in ("LEFT", "RIGHT"):
        mats += _mat(f"{side}.K", 3, 3, K) + _mat(f"{side}.D", 1, 5, [0] * 5) + _mat(f"{side}.R", 3, 3, eye) + \
            _mat(f"{side}.P", 3, 4, P)
    y = tmp_path / "cam.yaml"
    y.write_text(RECT_YAML.format(mats=mats))
    return str(mav) + "/", str(y), L, R


def test_euroc_input_conventions(tmp_path):
    """src/app/euroc_input.cpp:69-70,100-110: left <- cam1, right <- cam0, seconds since the first frame;
    rectification maps from LEFT.* / RIGHT.* (identity here: images unchanged)."""
    mav, y, L, R = _write_euroc(tmp_path)
    src = replay.EurocInput(mav, y)
    assert len(src) == 3
    for k in range(3):
        left, right, t = src.read(k)
        assert np.array_equal(left, L[k].numpy()) and np.array_equal(right, R[k].numpy())
        assert t == pytest.approx(0.05 * k, abs=1e-6)
    s = replay.read_settings(y)
    assert s["fx"] == 200.0 and s["search_x"] == 30


def test_rectification_maps():
    """cv::initUndistortRectifyMap: identity for K = P, R = I, D = 0; a principal point moved by 3 px
    shifts the source position by 3 px; remap with a half-pixel shift averages neighbours."""
    K = np.array([[200.0, 0, 160], [0, 200.0, 120], [0, 0, 1]])
    P = np.hstack([K, np.zeros((3, 1))])
    mx, my = replay.undistort_rectify_map(K, np.zeros(5), np.eye(3), P, (320, 240))
    u, v = np.meshgrid(np.arange(320, dtype=np.float32), np.arange(240, dtype=np.float32))
    assert np.allclose(mx, u, atol=1e-4) and np.allclose(my, v, atol=1e-4)
    K2 = K.copy(); K2[0, 2] += 3
    mx2, _ = replay.undistort_rectify_map(K2, np.zeros(5), np.eye(3), P, (320, 240))
    assert np.allclose(mx2, u + 3, atol=1e-4)
    img = (np.arange(240)[:, None] * 0 + np.arange(320)[None, :]).astype(np.uint8)
    out = replay.remap_linear(img, u + 0.5, v)
    assert np.array_equal(out[:, :250], np.rint(img[:, :250] + 0.5).astype(np.uint8))
    # barrel distortion pulls the corners inwards
    mx3, my3 = replay.undistort_rectify_map(K, np.array([-0.28, 0.07, 0, 0, 0]), np.eye(3), P, (320, 240))
    assert mx3[0, 0] > 0 and my3[0, 0] > 0 and abs(mx3[120, 160] - 160) < 1e-3


def test_side_by_side_input(tmp_path):
    """src/app/video_input.cpp:35-36: `right` is the left half of the frame, `left` the right half."""
    from PIL import Image
    frame = np.zeros((40, 120), np.uint8)
    frame[:, :60] = 10
    frame[:, 60:] = 200
    Image.fromarray(frame).save(str(tmp_path / "000000.png"))
    src = replay.SideBySideInput(str(tmp_path / "%06d.png"), 1, fps=25.0)
    left, right, t = src.read(0)
    assert left.shape == (40, 60) and np.all(left == 200) and np.all(right == 10) and t == pytest.approx(0.04)


def test_time_trace_uses_the_reference_stage_names():
    class St:
        stage_ms = [0.1, 0.01, 0.2, 0.3, 0.05, 0.07, 0.02, 0.4]
        is_keyframe = 1
    lines = replay.time_trace_lines(St)
    names = [l.split(" took: ")[0] for l in lines]
    assert names == ["Create pyramid", "estimator", "REFINEMENT: Optical flow", "pose refinement",
                     "Filter update", "Create new keyframe", "Stereo SLAM"]
    assert lines[3] == "pose refinement took: 0.3500ms"
    St.is_keyframe = 0
    assert "Create new keyframe" not in "".join(replay.time_trace_lines(St))


@pytest.mark.gpu
def test_replay_euroc_layout_equals_the_oracle(tmp_path, monkeypatch):
    """$SVO_DATA pointing at a EuRoC mav0/ directory: the harness reads it with the reference's
    conventions and the CSV it writes is the oracle's trajectory. No solver flag: the replay default
    is the library default, the reference-order Gauss-Newton."""
    import oracle_py as O
    import util
    mav, y, L, R = _write_euroc(tmp_path)
    monkeypatch.setenv("SVO_DATA", mav)
    out = tmp_path / "traj.csv"
    replay.main(["--settings", y, "--frames", "3", "-t", str(out)])
    rows = np.loadtxt(str(out), delimiter=",")
    cfg = dict(synth.CONFIGS["tiny"])
    ref = O.Slam(util.oracle_camera(cfg))
    for k in range(3):
        ref.new_image(L[k].numpy(), R[k].numpy(), float(np.float32(0.05 * k)))
        exp = np.concatenate([ref.pose()[:3], replay.csv_angles(ref.pose())])
        assert np.allclose(rows[k, 1:], exp, atol=1e-6), k


@pytest.mark.gpu
def test_replay_default_trace_is_the_oracles_and_time_trace_runs(capsys):
    """Replay() without flags tracks with the reference-order solver: pose and every GN trace equal
    the oracle's frame by frame. time_trace=True (the ctx does not exist before the first image)
    prints the reference's stage names from the first tracked frame on."""
    import oracle_py as O
    import util
    cfg, L, R, poses, ts = synth.make_sequence("tiny", 5, 4, device="cpu")
    ref = O.Slam(util.oracle_camera(cfg))
    rp = replay.Replay(cfg, time_trace=True)
    for k in range(5):
        ref.new_image(L[k].numpy(), R[k].numpy(), float(ts[k]))
        rp.feed(L[k].numpy(), R[k].numpy(), float(ts[k]))
        assert np.array_equal(rp.slam.get_frame().pose, ref.pose()), k
        if k:
            a, b = rp.slam.stats(), ref.stats()
            for lv in range(cfg["min_pyramid_level_pose_estimation"], cfg["max_pyramid_levels"]):
                x, y = a.sia_trace[lv], b.sia_trace[lv]
                assert (x.n_gradient, x.n_cost, x.n_accepted) == (y.n_gradient, y.n_cost, y.n_accepted), (k, lv)
            assert sum(a.stage_ms) > 0
    out = capsys.readouterr().out
    assert out.count("Stereo SLAM took:") == 5 and "estimator took:" in out


@pytest.mark.gpu
def test_replay_imu_loop_equals_the_oracle():
    """update_pose_from_imu (slam_app.cpp:111-135): the gyro samples that arrived since the last frame go
    through the pose filter before the next new_image — at most 104 * images_read / 30 of them, none
    before the first frame. The same calls on the oracle give the same poses, bit for bit."""
    import math
    import oracle_py as O
    import util
    cfg, L, R, poses, ts = synth.make_sequence("tiny", 5, 7, device="cpu")
    ref = O.Slam(util.oracle_camera(cfg))
    rp = replay.Replay(cfg)
    rng = np.random.RandomState(3)
    pv = np.full(6, 1000.0, np.float32)
    sv = np.array([100.0, 100.0, 100.0, 0.1, 0.1, 0.1], np.float32)
    for k in range(5):
        gyro = rng.normal(0, 2.0, (5, 3)).astype(np.float32)          # degrees per second; 5 arrive, 3 are used
        images_read = 1
        got = rp.update_pose_from_imu(gyro, images_read / 30.0) if k != 2 else None
        if k == 0:
            assert got is None                                        # no frame yet
        elif k != 2:
            n_used = min(len(gyro), int(np.float32(104.0) * np.float32(images_read / 30.0)))
            assert n_used == 3
            pose = ref.pose().astype(np.float32)
            for g in gyro[:n_used]:
                speed = np.zeros(6, np.float32)
                speed[3:] = (g.astype(np.float64) / 180.0 * math.pi).astype(np.float32)
                pose = ref.update_pose(pose, speed, pv, sv, 1.0 / 104.0)
            assert np.array_equal(got, pose), k
        if k == 2:                                                    # the same through feed()
            rp.feed(L[k].numpy(), R[k].numpy(), float(ts[k]), gyro_deg_s=gyro, images_read=2)
            pose = ref.pose().astype(np.float32)
            for g in gyro[:min(5, int(np.float32(104.0) * np.float32(2 / 30.0)))]:
                speed = np.zeros(6, np.float32)
                speed[3:] = (g.astype(np.float64) / 180.0 * math.pi).astype(np.float32)
                pose = ref.update_pose(pose, speed, pv, sv, 1.0 / 104.0)
            ref.new_image(L[k].numpy(), R[k].numpy(), float(ts[k]))
        else:
            rp.feed(L[k].numpy(), R[k].numpy(), float(ts[k]))
            ref.new_image(L[k].numpy(), R[k].numpy(), float(ts[k]))
        assert np.array_equal(rp.slam.get_frame().pose, ref.pose()), k
