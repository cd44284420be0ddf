"""ctypes binding of the CPU oracle (oracle/libsvo_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by the product package.  All arrays are
numpy host arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class CameraSettings(C.Structure):
    """svo_camera_settings (include/svo_types.h) == CameraSettings,
    src/include/stereo_slam_types.hpp:16-36."""
    _fields_ = [(n, C.c_float) for n in
                ("baseline", "fx", "fy", "cx", "cy", "k1", "k2", "k3", "p1", "p2")] + \
               [(n, C.c_int32) for n in
                ("grid_height", "grid_width", "search_x", "search_y",
                 "window_size_pose_estimator", "window_size_opt_flow",
                 "window_size_depth_calculator", "max_pyramid_levels",
                 "min_pyramid_level_pose_estimation")]


class Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("stride", C.c_int32)]


class GnTrace(C.Structure):
    _fields_ = [("level", C.c_int32), ("n_gradient", C.c_int32), ("n_cost", C.c_int32),
                ("n_accepted", C.c_int32), ("exit_small", C.c_int32),
                ("initial_cost", C.c_float), ("final_cost", C.c_float),
                ("pose", C.c_float * 6)]

    def as_dict(self):
        return dict(level=self.level, n_gradient=self.n_gradient, n_cost=self.n_cost,
                    n_accepted=self.n_accepted, exit_small=self.exit_small,
                    initial_cost=self.initial_cost, final_cost=self.final_cost,
                    pose=[float(v) for v in self.pose])


KP_INFO_DTYPE = np.dtype([
    ("score", "<f4"), ("level", "<i4"), ("type", "<i4"), ("keyframe_id", "<i4"),
    ("keypoint_index", "<i4"), ("color", "u1", (3,)), ("ignore_during_refinement", "u1"),
    ("ignore_completely", "u1"), ("ignore_temporary", "u1"), ("_pad", "u1", (2,)),
    ("outlier_count", "<i4"), ("inlier_count", "<i4"), ("kf_inv_depth", "<f4"),
    ("kf_variance", "<f4")], align=False)
assert KP_INFO_DTYPE.itemsize == 44


class FrameStats(C.Structure):
    _fields_ = [("n_tracked", C.c_int32), ("n_active", C.c_int32),
                ("sia_gradient_calls", C.c_int32), ("sia_cost_calls", C.c_int32),
                ("t_total", C.c_double), ("t_pyramid", C.c_double), ("t_sia", C.c_double),
                ("t_klt", C.c_double), ("t_reproj", C.c_double), ("t_disparity", C.c_double),
                ("t_filter", C.c_double), ("t_keyframe", C.c_double),
                ("pose_sia", C.c_float * 6), ("pose_refined", C.c_float * 6),
                ("sia_trace", GnTrace * 8), ("reproj_trace", GnTrace)]


def build(force=False):
    so = os.path.join(_HERE, "libsvo_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in
            ("cv_prims.c", "hot_path.c", "tracker.c", "svo_oracle.h", "oracle_internal.h")]
    if force or not os.path.exists(so) or \
            any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsvo_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libsvo_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.svo_o_sparse_align.restype = C.c_float
        _LIB.svo_o_reproj_gn.restype = C.c_float
        _LIB.svo_o_total_intensity_diff.restype = C.c_float
        _LIB.svo_o_slam_create.restype = C.c_void_p
    return _LIB


def make_camera(**kw):
    cam = CameraSettings()
    for k, v in kw.items():
        setattr(cam, k, v)
    return cam


def _img(a):
    assert a.dtype == np.uint8 and a.ndim == 2 and a.strides[1] == 1
    return Image(a.ctypes.data, a.shape[1], a.shape[0], a.strides[0])


def _imgs(arrs):
    arr = (Image * max(len(arrs), 1))()
    for i, a in enumerate(arrs):
        arr[i] = _img(a)
    return arr


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------- primitives
def rodrigues(r):
    r = _f32(r)
    R = np.zeros(9, np.float64)
    lib().svo_o_rodrigues(_p(r), _p(R))
    return R.reshape(3, 3)


def exponential_map(twist):
    t = _f32(twist)
    out = np.zeros(6, np.float32)
    lib().svo_o_exponential_map(_p(t), _p(out))
    return out


def inv_svd(A):
    A = _f32(A)
    n = A.shape[0]
    out = np.zeros((n, n), np.float32)
    ok = lib().svo_o_inv_svd(_p(A), n, _p(out))
    return out, ok


def solve_svd(A, b):
    A = _f32(A)
    b = _f32(b)
    x = np.zeros(A.shape[1], np.float32)
    lib().svo_o_solve_svd(_p(A), A.shape[0], A.shape[1], _p(b), _p(x))
    return x


def project_keypoints(pose, kps3d, cam):
    pose = _f32(pose)
    kps3d = _f32(kps3d)
    out = np.zeros((kps3d.shape[0], 2), np.float32)
    lib().svo_o_project_keypoints(_p(pose), _p(kps3d), kps3d.shape[0], C.byref(cam), _p(out))
    return out


def pyr_down(img):
    h, w = img.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().svo_o_pyr_down(_p(img), w, h, img.strides[0], _p(out), out.strides[0])
    return out


def scharr(img):
    h, w = img.shape
    out = np.zeros((h, w, 2), np.int16)
    lib().svo_o_scharr(_p(img), w, h, img.strides[0], _p(out))
    return out


def kf1_update(x, P, Q, R, meas):
    xx = C.c_float(x)
    PP = C.c_float(P)
    lib().svo_o_kf1_update(C.byref(xx), C.byref(PP), C.c_float(Q), C.c_float(R), C.c_float(meas))
    return xx.value, PP.value


# ------------------------------------------------------------------ hot path
def build_pyramid(img, n_levels):
    """halfSample pyramid; returns list of arrays (level 0 is `img`)."""
    img = np.ascontiguousarray(img)
    levels = [img]
    h, w = img.shape
    for _ in range(1, n_levels):
        h //= 2
        w //= 2
        levels.append(np.zeros((max(h, 0), max(w, 0)), np.uint8))
    arr = _imgs(levels)
    lib().svo_o_build_pyramid(C.byref(arr[0]), n_levels, arr)
    return levels


def build_lk_pyramid(img, win, max_levels=3):
    img = np.ascontiguousarray(img)
    levels = [img]
    h, w = img.shape
    for _ in range(1, max_levels):
        h = (h + 1) // 2
        w = (w + 1) // 2
        levels.append(np.zeros((h, w), np.uint8))
    arr = _imgs(levels)
    n = lib().svo_o_build_lk_pyramid(C.byref(arr[0]), max_levels, win, arr)
    return levels[:n]


def total_intensity_diff(img1, img2, kps1, kps2, patch=4):
    kps1 = _f32(kps1)
    kps2 = _f32(kps2)
    a, b = _img(img1), _img(img2)
    return lib().svo_o_total_intensity_diff(C.byref(a), C.byref(b), _p(kps1), _p(kps2),
                                            kps1.shape[0], patch)


def sparse_align(prev_pyr, cur_pyr, kps2d, kps3d, flags, cam, pose_guess):
    kps2d = _f32(kps2d)
    kps3d = _f32(kps3d)
    flags = np.ascontiguousarray(flags, dtype=np.uint32)
    guess = _f32(pose_guess)
    out = np.zeros(6, np.float32)
    trace = (GnTrace * 8)()
    pp, cp = _imgs(prev_pyr), _imgs(cur_pyr)
    cost = lib().svo_o_sparse_align(pp, cp, _p(kps2d), _p(kps3d), _p(flags), kps2d.shape[0],
                                    C.byref(cam), _p(guess), _p(out), trace)
    return out, cost, [trace[i].as_dict() for i in range(cam.max_pyramid_levels)]


def sia_gradient(prev, cur, level, kps2d, kps3d, flags, cam, pose):
    kps2d = _f32(kps2d)
    kps3d = _f32(kps3d)
    flags = np.ascontiguousarray(flags, dtype=np.uint32)
    pose = _f32(pose)
    H = np.zeros(36, np.float32)
    b = np.zeros(6, np.float32)
    step = np.zeros(6, np.float32)
    a, c = _img(prev), _img(cur)
    lib().svo_o_sia_gradient(C.byref(a), C.byref(c), level, _p(kps2d), _p(kps3d), _p(flags),
                             kps2d.shape[0], C.byref(cam), _p(pose), _p(H), _p(b), _p(step))
    return H.reshape(6, 6), b, step


def klt_track(prev_lk, cur_lk, prev_pts, cur_pts, win):
    prev_pts = _f32(prev_pts)
    cur = _f32(cur_pts).copy()
    n = prev_pts.shape[0]
    status = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    nl = min(len(prev_lk), len(cur_lk))
    lib().svo_o_klt_track(_imgs(prev_lk), _imgs(cur_lk), nl, _p(prev_pts), _p(cur), n, win,
                          _p(status), _p(err))
    return cur, status, err


def refine_merge(kps2d, flags, tracked, err):
    kps2d = _f32(kps2d).copy()
    flags = np.ascontiguousarray(flags, dtype=np.uint32).copy()
    tracked = _f32(tracked)
    err = _f32(err)
    lib().svo_o_refine_merge(_p(kps2d), _p(flags), _p(tracked), _p(err), kps2d.shape[0])
    return kps2d, flags


def reproj_gn(kps2d, kps3d, flags, cam, pose_in):
    kps2d = _f32(kps2d)
    kps3d = _f32(kps3d)
    flags = np.ascontiguousarray(flags, dtype=np.uint32)
    pose_in = _f32(pose_in)
    out = np.zeros(6, np.float32)
    tr = GnTrace()
    cost = lib().svo_o_reproj_gn(_p(kps2d), _p(kps3d), _p(flags), kps2d.shape[0], C.byref(cam),
                                 _p(pose_in), _p(out), C.byref(tr))
    return out, cost, tr.as_dict()


def ssd_disparity(left, right, kps2d, win, search_x, search_y, clamp_half=1):
    kps2d = _f32(kps2d)
    n = kps2d.shape[0]
    out = np.zeros(n, np.float32)
    a, b = _img(left), _img(right)
    lib().svo_o_ssd_disparity(C.byref(a), C.byref(b), _p(kps2d), n, win, search_x, search_y,
                              clamp_half, _p(out))
    return out


def outlier_check(kps2d, disparity, cam, frame_pose, ref3d, kf_pose, outlier, inlier):
    kps2d = _f32(kps2d)
    disparity = _f32(disparity)
    frame_pose = _f32(frame_pose)
    ref3d = _f32(ref3d)
    kf_pose = _f32(kf_pose)
    outlier = np.ascontiguousarray(outlier, dtype=np.int32).copy()
    inlier = np.ascontiguousarray(inlier, dtype=np.int32).copy()
    lib().svo_o_outlier_check(_p(kps2d), _p(disparity), kps2d.shape[0], C.byref(cam),
                              _p(frame_pose), _p(ref3d), _p(kf_pose), _p(outlier), _p(inlier))
    return outlier, inlier


def update_kps3d(kps2d, kps3d, flags, cam, frame_pose, ref2d, kf_pose, outlier, kf_x, kf_p):
    kps2d = _f32(kps2d)
    kps3d = _f32(kps3d).copy()
    flags = np.ascontiguousarray(flags, dtype=np.uint32)
    frame_pose = _f32(frame_pose)
    ref2d = _f32(ref2d)
    kf_pose = _f32(kf_pose)
    outlier = np.ascontiguousarray(outlier, dtype=np.int32).copy()
    kf_x = _f32(kf_x).copy()
    kf_p = _f32(kf_p).copy()
    lib().svo_o_update_kps3d(_p(kps2d), _p(kps3d), _p(flags), kps2d.shape[0], C.byref(cam),
                             _p(frame_pose), _p(ref2d), _p(kf_pose), _p(outlier), _p(kf_x), _p(kf_p))
    return kps3d, outlier, kf_x, kf_p


def fast_score_nms(img, threshold=6):
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().svo_o_fast_score_nms(_p(img), w, h, img.strides[0], threshold, _p(out))
    return out


def sobel_x_u8(img):
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().svo_o_sobel_x_u8(_p(img), w, h, img.strides[0], _p(out))
    return out


def detect_keypoints(img, grid_w, grid_h, level=0):
    h, w = img.shape
    cap = (w // grid_w + 1) * (h // grid_h + 1) + 1
    kps = np.zeros((cap, 2), np.float32)
    score = np.zeros(cap, np.float32)
    typ = np.zeros(cap, np.int32)
    n = lib().svo_o_detect_keypoints(_p(img), w, h, img.strides[0], grid_w, grid_h, level,
                                     _p(kps), _p(score), _p(typ), cap)
    return kps[:n], score[:n], typ[:n]


# ------------------------------------------------------------------- tracker
class Slam:
    """Oracle restatement of StereoSlam (src/include/stereo_slam.hpp:27-79)."""

    def __init__(self, cam):
        self._cam = cam
        self._h = C.c_void_p(lib().svo_o_slam_create(C.byref(cam)))

    def close(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.svo_o_slam_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def new_image(self, left, right, time_stamp):
        left = np.ascontiguousarray(left)
        right = np.ascontiguousarray(right)
        h, w = left.shape
        return lib().svo_o_slam_new_image(self._h, _p(left), _p(right), w, h, C.c_float(time_stamp))

    def pose(self):
        p = np.zeros(6, np.float32)
        lib().svo_o_slam_get_pose(self._h, _p(p))
        return p

    def num_keyframes(self):
        return lib().svo_o_slam_num_keyframes(self._h)

    def keypoints(self):
        n = lib().svo_o_slam_num_keypoints(self._h)
        k2 = np.zeros((n, 2), np.float32)
        k3 = np.zeros((n, 3), np.float32)
        info = np.zeros(n, KP_INFO_DTYPE)
        lib().svo_o_slam_get_keypoints(self._h, _p(k2), _p(k3), _p(info), n)
        return k2, k3, info

    def keyframe(self, kid):
        n = lib().svo_o_slam_get_keyframe_keypoints(self._h, kid, None, None, None, None, 0)
        k2 = np.zeros((n, 2), np.float32)
        k3 = np.zeros((n, 3), np.float32)
        info = np.zeros(n, KP_INFO_DTYPE)
        pose = np.zeros(6, np.float32)
        lib().svo_o_slam_get_keyframe_keypoints(self._h, kid, _p(k2), _p(k3), _p(info), _p(pose), n)
        return k2, k3, info, pose

    def stats(self):
        st = FrameStats()
        lib().svo_o_slam_get_stats(self._h, C.byref(st))
        return st

    def update_pose(self, pose, speed, pose_var, speed_var, dt):
        out = np.zeros(6, np.float32)
        a, b, c, d = _f32(pose), _f32(speed), _f32(pose_var), _f32(speed_var)
        lib().svo_o_slam_update_pose(self._h, _p(a), _p(b), _p(c), _p(d), C.c_double(dt), _p(out))
        return out
