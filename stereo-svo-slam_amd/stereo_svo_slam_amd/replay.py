"""Headless replay harness: the counterpart of SlamApp::process_image
(src/app/slam_app.cpp:160-196), of the trajectory CSV writer in SlamApp::stop
(:229-244), of the YAML reader ImageInput::read_settings (src/app/image_input.cpp:13-37)
and of the evaluation scripts test/extract_fps.py:26-34 / test/extract_extremas.py:31-54.

Host bookkeeping only; every frame goes through StereoSlam.new_image (libsvo_hip.so).

    python -m stereo_svo_slam_amd.replay --synthetic euroc --frames 100 -t traj.csv
    python -m stereo_svo_slam_amd.replay --settings EuRoC.yaml --pairs 'seq/%06d_left.png,seq/%06d_right.png' -t traj.csv
"""
import argparse
import math
import re
import time

import numpy as np

from . import synth
from .stereo_slam import StereoSlam

# key in the YAML -> CameraSettings field (src/app/image_input.cpp:16-36)
_YAML_KEYS = {
    "Camera1.fx": "fx", "Camera1.fy": "fy", "Camera1.cx": "cx", "Camera1.cy": "cy",
    "Camera.baseline": "baseline",
    "Camera.window_size_pose_estimator": "window_size_pose_estimator",
    "Camera.window_size_opt_flow": "window_size_opt_flow",
    "Camera.window_size_depth_calculator": "window_size_depth_calculator",
    "Camera.max_pyramid_levels": "max_pyramid_levels",
    "Camera.min_pyramid_level_pose_estimation": "min_pyramid_level_pose_estimation",
    "Camera1.k1": "k1", "Camera1.k2": "k2", "Camera1.k3": "k3", "Camera1.p1": "p1", "Camera1.p2": "p2",
    "Camera.grid_width": "grid_width", "Camera.grid_height": "grid_height",
    "Camera.search_x": "search_x", "Camera.search_y": "search_y",
}
_INT_FIELDS = {"window_size_pose_estimator", "window_size_opt_flow", "window_size_depth_calculator",
               "max_pyramid_levels", "min_pyramid_level_pose_estimation", "grid_width", "grid_height",
               "search_x", "search_y"}


def read_settings(path):
    """cv::FileStorage YAML (`%YAML:1.0` header, `key: value # comment` scalars; matrices are
    skipped) -> dict with the CameraSettings fields. Missing keys read as 0 like cv::FileNode."""
    out = {f: (0 if f in _INT_FIELDS else 0.0) for f in _YAML_KEYS.values()}
    with open(path) as fh:
        for line in fh:
            m = re.match(r"^([A-Za-z0-9_.]+)\s*:\s*([-+0-9.eE]+)\s*(#.*)?$", line.strip())
            if m and m.group(1) in _YAML_KEYS:
                f = _YAML_KEYS[m.group(1)]
                out[f] = int(float(m.group(2))) if f in _INT_FIELDS else float(m.group(2))
            m2 = re.match(r"^(Camera\.(width|height|image_width|image_height))\s*:\s*(\d+)", line.strip())
            if m2:
                out["width" if "width" in m2.group(1) else "height"] = int(m2.group(3))
    return out


def _rodrigues(r):
    r = np.asarray(r, np.float64)
    th = np.linalg.norm(r)
    if th < 1e-15:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return math.cos(th) * np.eye(3) + (1 - math.cos(th)) * np.outer(k, k) + math.sin(th) * K


def _rodrigues_inv(R):
    """Rotation matrix -> axis-angle vector (cv::Rodrigues, matrix input)."""
    c = min(1.0, max(-1.0, (np.trace(R) - 1) / 2))
    th = math.acos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(v) / 2
    if s < 1e-10:
        if c > 0:
            return np.zeros(3)
        # theta = pi: axis from the diagonal
        a = np.sqrt(np.maximum((np.diag(R) + 1) / 2, 0))
        if R[0, 1] < 0:
            a[1] = -a[1]
        if R[0, 2] < 0:
            a[2] = -a[2]
        return a / max(np.linalg.norm(a), 1e-30) * th
    return v / (2 * s) * th


def csv_angles(pose):
    """Angles written to the trajectory CSV: Rodrigues((R_y R_x) R_z), slam_app.cpp:232-241."""
    Rx = _rodrigues([pose[3], 0, 0])
    Ry = _rodrigues([0, pose[4], 0])
    Rz = _rodrigues([0, 0, pose[5]])
    return _rodrigues_inv((Ry @ Rx) @ Rz)


def write_trajectory_csv(path, cumulative_times, trajectory):
    """`t_cumulative_algorithm_seconds,x,y,z,rx',ry',rz'` per frame (slam_app.cpp:229-244)."""
    with open(path, "w") as fh:
        for t, pose in zip(cumulative_times, trajectory):
            a = csv_angles(pose)
            fh.write(",".join(f"{v:.6g}" for v in (t, pose[0], pose[1], pose[2], a[0], a[1], a[2])) + "\n")


def fps_from_csv_rows(rows):
    """test/extract_fps.py:26-34: n / (t_last - t_first) over the cumulative algorithm time."""
    rows = np.asarray(rows, np.float64)
    dt = rows[-1, 0] - rows[0, 0]
    return rows.shape[0] / dt if dt > 0 else float("nan")


def error_report(test_rows, reference_rows):
    """test/extract_extremas.py:31-54: max / mean absolute error (angles in degrees) and FPS."""
    t = np.asarray(test_rows, np.float64)
    r = np.asarray(reference_rows, np.float64)[: t.shape[0]]
    d = np.abs(t - r)
    d[:, 4:] = d[:, 4:] / math.pi * 180
    return dict(max=d.max(0)[1:].tolist(), mean=d.mean(0)[1:].tolist(), fps=fps_from_csv_rows(t))


class Replay:
    """process_image loop: only the time inside new_image is accumulated (slam_app.cpp:186-190)."""

    def __init__(self, settings, device=0):
        self.settings = settings
        self.slam = StereoSlam(settings, device=device)
        self.cumulative = []
        self._t = 0.0

    def feed(self, left, right, time_stamp):
        t0 = time.perf_counter()
        self.slam.new_image(left, right, time_stamp)
        self._t += time.perf_counter() - t0
        self.cumulative.append(self._t)

    def rows(self):
        traj = self.slam.get_trajectory()
        return np.array([[t, p[0], p[1], p[2], *csv_angles(p)] for t, p in zip(self.cumulative, traj)])

    def write(self, path):
        write_trajectory_csv(path, self.cumulative, self.slam.get_trajectory())


def _load_pair(pattern, k):
    from PIL import Image
    lp, rp = pattern.split(",")
    load = lambda p: np.ascontiguousarray(np.array(Image.open(p % k).convert("L")))
    return load(lp), load(rp)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--settings", help="cv::FileStorage YAML with the Camera.* keys")
    ap.add_argument("--synthetic", choices=sorted(synth.CONFIGS), help="seeded synthetic sequence")
    ap.add_argument("--pairs", help="'left_%%06d.png,right_%%06d.png': the library's left / right images")
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--rate", type=float, default=20.0, help="frames per second of the time stamps")
    ap.add_argument("-t", "--trajectory", help="output CSV")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)

    gt = None
    if args.synthetic:
        cfg, L, R, gt, ts = synth.make_sequence(args.synthetic, args.frames, args.seed, device="cpu")
        settings = read_settings(args.settings) if args.settings else cfg
        frames = ((L[k].numpy(), R[k].numpy(), float(ts[k])) for k in range(args.frames))
    elif args.pairs and args.settings:
        settings = read_settings(args.settings)
        frames = ((*_load_pair(args.pairs, k), k / args.rate) for k in range(args.frames))
    else:
        ap.error("give --synthetic or --settings with --pairs")
    rp = Replay(settings, args.device)
    for left, right, t in frames:
        rp.feed(left, right, t)
    rows = rp.rows()
    print(f"frames {rows.shape[0]}  Average FPS: {fps_from_csv_rows(rows):.2f}  "
          f"keyframes {rp.slam.num_keyframes()}")
    if gt is not None:
        ref = np.array([[0, p[0], p[1], p[2], *csv_angles(p)] for p in gt])
        ref[:, 0] = rows[:, 0]
        rep = error_report(rows, ref)
        print("max  |err| x y z [m] rx ry rz [deg]:", " ".join(f"{v:.3f}" for v in rep["max"]))
        print("mean |err|                          :", " ".join(f"{v:.3f}" for v in rep["mean"]))
    if args.trajectory:
        rp.write(args.trajectory)


if __name__ == "__main__":
    main()
