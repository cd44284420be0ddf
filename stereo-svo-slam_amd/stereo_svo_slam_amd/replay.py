"""Headless replay harness: the counterpart of SlamApp::process_image
(src/app/slam_app.cpp:160-196), of the trajectory CSV writer in SlamApp::stop
(:229-244), of the YAML reader ImageInput::read_settings (src/app/image_input.cpp:13-37)
and of the evaluation scripts test/extract_fps.py:26-34 / test/extract_extremas.py:31-54.

Host bookkeeping only; every frame goes through StereoSlam.new_image (libsvo_hip.so).

    python -m stereo_svo_slam_amd.replay --synthetic euroc --frames 100 -t traj.csv
    python -m stereo_svo_slam_amd.replay --settings EuRoC.yaml --euroc /data/MH_02_easy/mav0/ -t traj.csv
    python -m stereo_svo_slam_amd.replay --settings Blender.yaml --sbs 'frames/%06d.png' -t traj.csv
    python -m stereo_svo_slam_amd.replay --settings EuRoC.yaml --pairs 'seq/%06d_left.png,seq/%06d_right.png'

Inputs follow the reference's conventions (the library's `left` is the physically RIGHT camera):
EurocInput (src/app/euroc_input.cpp:48-70,100-105), VideoInput (src/app/video_input.cpp:35-36).
With $SVO_DATA set (a EuRoC `mav0/` directory, or a directory of side-by-side frames) and no
explicit input that data is used; otherwise the seeded synthetic sequence.
"""
import argparse
import math
import os
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


def read_matrix(path, key):
    """A `key: !!opencv-matrix` entry of a cv::FileStorage YAML (rows, cols, data) as float64 array,
    or None (LEFT.K, LEFT.D, LEFT.R, LEFT.P ... of src/app/EuRoC.yaml)."""
    txt = open(path).read()
    m = re.search(re.escape(key) + r"\s*:\s*!!opencv-matrix\s*rows:\s*(\d+)\s*cols:\s*(\d+)\s*dt:\s*\w+\s*data:\s*\[([^\]]*)\]",
                  txt, re.S)
    if not m:
        return None
    return np.array([float(v) for v in m.group(3).replace("\n", " ").split(",")], np.float64).reshape(
        int(m.group(1)), int(m.group(2)))


def read_scalar(path, key, default=0):
    m = re.search(r"^" + re.escape(key) + r"\s*:\s*([-+0-9.eE]+)", open(path).read(), re.M)
    return float(m.group(1)) if m else default


def undistort_rectify_map(K, D, R, P, size):
    """cv::initUndistortRectifyMap(K, D, R, P[:3,:3], size, CV_32F): for every pixel of the rectified
    image the position in the raw image (map_x, map_y), 5-coefficient Brown model."""
    w, h = size
    iR = np.linalg.inv(np.asarray(P, np.float64)[:3, :3] @ np.asarray(R, np.float64))
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    pts = np.stack([u, v, np.ones_like(u)], -1) @ iR.T
    x, y = pts[..., 0] / pts[..., 2], pts[..., 1] / pts[..., 2]
    d = list(np.ravel(D)) + [0.0] * 5
    k1, k2, p1, p2, k3 = d[:5]
    r2 = x * x + y * y
    kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * kr + p1 * 2 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * kr + p1 * (r2 + 2 * y * y) + p2 * 2 * x * y
    K = np.asarray(K, np.float64)
    return (K[0, 0] * xd + K[0, 2]).astype(np.float32), (K[1, 1] * yd + K[1, 2]).astype(np.float32)


def remap_linear(img, map_x, map_y):
    """cv::remap(img, ., map_x, map_y, INTER_LINEAR) with a constant 0 border, 8-bit."""
    h, w = img.shape
    x0 = np.floor(map_x).astype(np.int64)
    y0 = np.floor(map_y).astype(np.int64)
    ax, ay = map_x - x0, map_y - y0
    src = np.pad(img.astype(np.float32), 1)

    def at(yy, xx):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        return np.where(ok, src[np.clip(yy, -1, h) + 1, np.clip(xx, -1, w) + 1], 0.0)

    v = (at(y0, x0) * (1 - ax) * (1 - ay) + at(y0, x0 + 1) * ax * (1 - ay) +
         at(y0 + 1, x0) * (1 - ax) * ay + at(y0 + 1, x0 + 1) * ax * ay)
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def _gray(path):
    from PIL import Image
    return np.ascontiguousarray(np.array(Image.open(path).convert("L")))


class EurocInput:
    """EurocInput (src/app/euroc_input.cpp): `image_path` is the mav0/ directory. cam0/data.csv lists
    time stamps [ns] and file names; the library's `right` image is cam0 rectified with the LEFT.*
    calibration of the settings file, `left` is cam1 rectified with RIGHT.* (:69-70, :100-101); time
    stamps are seconds since the first frame as float (:104-110)."""

    def __init__(self, image_path, settings):
        self.right_images, self.left_images, self.timestamps = [], [], []
        t0 = None
        with open(os.path.join(image_path, "cam0", "data.csv")) as fh:
            for line in fh:
                line = line.strip()
                if not line or line.startswith("#"):
                    continue
                stamp, name = line.split(",")[0], line.split(",")[-1].strip()
                self.right_images.append(os.path.join(image_path, "cam0", "data", name))
                self.left_images.append(os.path.join(image_path, "cam1", "data", name))
                t = float(stamp) / 1.0e9
                t0 = t if t0 is None else t0
                self.timestamps.append(np.float32(t - t0))
        self.maps_l = self.maps_r = None
        mats = {k: read_matrix(settings, k) for k in ("LEFT.K", "LEFT.D", "LEFT.R", "LEFT.P",
                                                     "RIGHT.K", "RIGHT.D", "RIGHT.R", "RIGHT.P")}
        if all(v is not None for v in mats.values()):
            size_l = (int(read_scalar(settings, "LEFT.width")), int(read_scalar(settings, "LEFT.height")))
            size_r = (int(read_scalar(settings, "RIGHT.width")), int(read_scalar(settings, "RIGHT.height")))
            self.maps_l = undistort_rectify_map(mats["LEFT.K"], mats["LEFT.D"], mats["LEFT.R"], mats["LEFT.P"], size_l)
            self.maps_r = undistort_rectify_map(mats["RIGHT.K"], mats["RIGHT.D"], mats["RIGHT.R"], mats["RIGHT.P"], size_r)

    def __len__(self):
        return len(self.left_images)

    def read(self, k):
        """(left, right, time_stamp) in the library's naming."""
        cam0, cam1 = _gray(self.right_images[k]), _gray(self.left_images[k])
        if self.maps_l is not None:
            cam0 = remap_linear(cam0, *self.maps_l)      # right <- remap(cam0, M1l, M2l)
            cam1 = remap_linear(cam1, *self.maps_r)      # left  <- remap(cam1, M1r, M2r)
        return cam1, cam0, float(self.timestamps[k])


class SideBySideInput:
    """VideoInput (src/app/video_input.cpp:25-45) on decoded frames: every image holds both cameras
    side by side; `right` is the LEFT half, `left` the RIGHT half; time stamps advance by 1/fps from
    1/fps."""

    def __init__(self, pattern, n_frames, fps=30.0):
        self.pattern, self.n, self.fps = pattern, n_frames, fps

    def __len__(self):
        return self.n

    def read(self, k):
        img = _gray(self.pattern % k)
        w = img.shape[1] // 2
        right = np.ascontiguousarray(img[:, :w])
        left = np.ascontiguousarray(img[:, w:2 * w])
        return left, right, (k + 1) / self.fps


# END_MEASUREMENT names of the reference (src/lib/stereo_slam.cpp:66-86,140,227,235;
# src/lib/pose_refinement.cpp:120) over the stage times of svo_frame_stats (HIP events)
def time_trace_lines(stats):
    st = list(stats.stage_ms)
    lines = [("Create pyramid", st[0]), ("estimator", st[2]), ("REFINEMENT: Optical flow", st[3]),
             ("pose refinement", st[3] + st[4]), ("Filter update", st[5] + st[6])]
    if stats.is_keyframe:
        lines.append(("Create new keyframe", st[7]))
    lines.append(("Stereo SLAM", sum(st)))
    return [f"{name} took: {ms:.4f}ms" for name, ms in lines]


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

    def __init__(self, settings, device=0, time_trace=False, fast=False):
        """fast=False keeps the library default: the reference-order Gauss-Newton (bit-exact traces)."""
        self.settings = settings
        self.slam = StereoSlam(settings, device=device)
        if fast:
            self.slam.set_fast_solver(True)
        self.cumulative = []
        self._t = 0.0
        self.time_trace = time_trace
        if time_trace:
            self.slam.enable_timing(True)

    IMU_RATE = 104.0                              # samples per second the Econ camera's IMU thread delivers

    def update_pose_from_imu(self, gyro_deg_s, dt):
        """SlamApp::update_pose_from_imu (slam_app.cpp:111-135): between two frames the pose filter is fed
        the gyro rates of at most IMU_RATE * dt samples ([n, 3] degrees per second, x y z) as the speed
        measurement with the app's variances, one StereoSlam::update_pose call of 1 / IMU_RATE each;
        nothing happens before the first frame. Returns the filtered pose (the app discards it: the calls
        act through the filter's state)."""
        if self.slam._ctx is None:                # (the ctx is created by the first frame)
            return None
        gyro = np.asarray(gyro_deg_s, np.float32).reshape(-1, 3)
        pose_variance = np.full(6, 1000.0, np.float32)
        speed_variance = np.array([100.0, 100.0, 100.0, 0.1, 0.1, 0.1], np.float32)
        pose = np.asarray(self.slam.pose(), np.float32)
        n = min(len(gyro), int(np.float32(self.IMU_RATE) * np.float32(dt)))    # std::min<size_t>(size, f * dt): truncated
        for g in gyro[:n]:
            speed = np.zeros(6, np.float32)
            speed[3:] = (g.astype(np.float64) / 180.0 * math.pi).astype(np.float32)
            pose = self.slam.update_pose(pose, speed, pose_variance, speed_variance, 1.0 / self.IMU_RATE)   # double 1.0 / f
        return pose

    def feed(self, left, right, time_stamp, gyro_deg_s=None, images_read=1):
        """process_image (slam_app.cpp:160-196); with IMU samples: update_pose_from_imu(images_read / 30) first"""
        if gyro_deg_s is not None:
            self.update_pose_from_imu(gyro_deg_s, images_read / 30.0)
        t0 = time.perf_counter()
        self.slam.new_image(left, right, time_stamp)
        self._t += time.perf_counter() - t0
        self.cumulative.append(self._t)
        if self.time_trace:                       # like PRINT_TIME_TRACE of the reference
            print("\n".join(time_trace_lines(self.slam.stats())))

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
    ap.add_argument("--euroc", help="EuRoC mav0/ directory (cam0, cam1): EurocInput conventions")
    ap.add_argument("--sbs", help="'frames/%%06d.png' side-by-side frames: VideoInput conventions")
    ap.add_argument("--time-trace", action="store_true", help="print '<stage> took: X ms' per frame (reference names)")
    ap.add_argument("--fast", action="store_true",
                    help="svo_ctx_set_fast_solver(1): tree sums + LDL^T instead of the default reference-order Gauss-Newton")
    ap.add_argument("--exact", action="store_true", help="no-op (the reference-order mode is the default)")
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--rate", type=float, default=20.0, help="frames per second of the time stamps")
    ap.add_argument("-t", "--trajectory", help="output CSV")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)

    gt = None
    data = os.environ.get("SVO_DATA")
    if data and not (args.synthetic or args.pairs or args.euroc or args.sbs):
        if os.path.exists(os.path.join(data, "cam0", "data.csv")):
            args.euroc = data
        else:
            args.sbs = os.path.join(data, "%06d.png")
    if args.euroc and args.settings:
        settings = read_settings(args.settings)
        src = EurocInput(args.euroc, args.settings)
        n = min(args.frames, len(src))
        frames = (src.read(k) for k in range(n))
    elif args.sbs and args.settings:
        settings = read_settings(args.settings)
        src = SideBySideInput(args.sbs, args.frames, args.rate)
        frames = (src.read(k) for k in range(args.frames))
    elif args.synthetic:
        cfg, L, R, gt, ts = synth.make_sequence(args.synthetic, args.frames, args.seed, device="cpu")
        settings = read_settings(args.settings) if args.settings else cfg
        frames = ((L[k].numpy(), R[k].numpy(), float(ts[k])) for k in range(args.frames))
    elif args.pairs and args.settings:
        settings = read_settings(args.settings)
        frames = ((*_load_pair(args.pairs, k), k / args.rate) for k in range(args.frames))
    else:
        ap.error("give --synthetic, or --settings with --euroc / --sbs / --pairs (or $SVO_DATA)")
    rp = Replay(settings, args.device, args.time_trace, args.fast)
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
