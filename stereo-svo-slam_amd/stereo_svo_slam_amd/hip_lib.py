"""ctypes binding of the product library libsvo_hip.so (C ABI: include/svo_hip.h).

torch is used only as plumbing: device buffers and the current HIP stream.
There is no CPU fallback — a missing library or GPU raises.
"""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
LIB_PATH = os.environ.get("SVO_HIP_LIB", os.path.join(CSRC, "libsvo_hip.so"))   # override: diagnostic builds
_LIB = None

# every symbol include/svo_hip.h declares
SYMBOLS = (
    "svo_last_error", "svo_version", "svo_handle_create", "svo_handle_destroy",
    "svo_handle_set_stream", "svo_handle_synchronize", "svo_build_pyramid",
    "svo_build_lk_pyramid", "svo_sparse_align", "svo_klt_track", "svo_reproj_gn",
    "svo_ssd_disparity", "svo_depth_filter_update",
    "svo_ctx_create", "svo_ctx_destroy", "svo_new_images", "svo_submit_images", "svo_wait",
    "svo_ctx_get_groups", "svo_new_image", "svo_get_pose",
    "svo_get_frame_keypoints", "svo_get_keyframe_count", "svo_get_keyframe",
    "svo_get_trajectory", "svo_update_pose", "svo_get_frame_stats", "svo_ctx_enable_timing",
    "svo_get_totals", "svo_handle_set_exact_pinv", "svo_ctx_set_exact_pinv",
    "svo_handle_set_fast_solver", "svo_ctx_set_fast_solver",
    "svo_device_malloc", "svo_device_free", "svo_copy_to_device", "svo_copy_to_host",
    "svo_copy_image_to_device", "svo_project_keypoints",
)


class SvoError(RuntimeError):
    pass


class CameraSettings(C.Structure):
    """svo_camera_settings == CameraSettings, src/include/stereo_slam_types.hpp:16-36."""
    _fields_ = [(n, C.c_float) for n in
                ("baseline", "fx", "fy", "cx", "cy", "k1", "k2", "k3", "p1", "p2")] + \
               [(n, C.c_int32) for n in
                ("grid_height", "grid_width", "search_x", "search_y",
                 "window_size_pose_estimator", "window_size_opt_flow",
                 "window_size_depth_calculator", "max_pyramid_levels",
                 "min_pyramid_level_pose_estimation")]

    @classmethod
    def from_dict(cls, d):
        cam = cls()
        for name, _ in cls._fields_:
            setattr(cam, name, d[name])
        return cam


class Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("stride", C.c_int32)]


GN_TRACE_DTYPE = np.dtype([("level", "<i4"), ("n_gradient", "<i4"), ("n_cost", "<i4"),
                           ("n_accepted", "<i4"), ("exit_small", "<i4"),
                           ("initial_cost", "<f4"), ("final_cost", "<f4"), ("pose", "<f4", (6,))])
assert GN_TRACE_DTYPE.itemsize == 52


def lib():
    """Load libsvo_hip.so; fail loudly when it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SvoError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _LIB = C.CDLL(LIB_PATH)
        _LIB.svo_last_error.restype = C.c_char_p
    return _LIB


def _check(rc):
    if rc != 0:
        raise SvoError(f"libsvo_hip error {rc}: {lib().svo_last_error().decode()}")


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device, contiguous tensors only"
    return C.c_void_p(t.data_ptr())


def _img(t):
    assert t.dtype == torch.uint8 and t.dim() == 2 and t.is_cuda and t.stride(1) == 1
    return Image(t.data_ptr(), t.shape[1], t.shape[0], t.stride(0))


def _imgs(ts, n=None):
    arr = (Image * max(n or len(ts), 1))()
    for i, t in enumerate(ts):
        arr[i] = _img(t)
    return arr


class Handle:
    """svo_handle: one per (GPU, caller)."""

    def __init__(self, device=0, max_keypoints=4096):
        if not torch.cuda.is_available():
            raise SvoError("no GPU visible: libsvo_hip has no CPU fallback")
        self.device = torch.device("cuda", device)
        self._h = C.c_void_p()
        _check(lib().svo_handle_create(device, max_keypoints, C.byref(self._h)))
        self.use_current_stream()

    def use_current_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        _check(lib().svo_handle_set_stream(self._h, C.c_void_p(s)))

    def synchronize(self):
        _check(lib().svo_handle_synchronize(self._h))

    def set_fast_solver(self, on=True):
        """svo_handle_set_fast_solver: off (default) = the reference's row-by-row normal equations +
        SVD pseudo-inverse; on = tree sums + LDL^T."""
        _check(lib().svo_handle_set_fast_solver(self._h, int(on)))

    def set_exact_pinv(self, on=True):
        self.set_fast_solver(not on)

    def close(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.svo_handle_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- P1 ---------------------------------------------------------------
    def build_pyramid(self, img, n_levels):
        """createImgPyramid (src/lib/stereo_slam.cpp:112-121): list of uint8 device tensors."""
        levels = [img]
        h, w = img.shape
        for _ in range(1, n_levels):
            h //= 2
            w //= 2
            levels.append(torch.empty((h, w), dtype=torch.uint8, device=img.device))
        arr = _imgs(levels)
        _check(lib().svo_build_pyramid(self._h, n_levels, arr))
        return levels

    # -- P2 ---------------------------------------------------------------
    def build_lk_pyramid(self, img, win, max_levels=3):
        levels = [img]
        h, w = img.shape
        for _ in range(1, max_levels):
            h = (h + 1) // 2
            w = (w + 1) // 2
            levels.append(torch.empty((h, w), dtype=torch.uint8, device=img.device))
        arr = _imgs(levels)
        n = C.c_int(0)
        _check(lib().svo_build_lk_pyramid(self._h, max_levels, win, arr, C.byref(n)))
        return levels[:n.value]

    # -- A ----------------------------------------------------------------
    def sparse_align(self, prev_pyr, cur_pyr, kps2d, kps3d, flags, cam, pose_guess,
                     dbg_level=-1):
        """PoseEstimator::estimate_pose. Returns (pose[6], cost, trace, dbg) as device/np."""
        dev = kps2d.device
        n = kps2d.shape[0]
        pose_out = torch.zeros(6, dtype=torch.float32, device=dev)
        cost = torch.zeros(1, dtype=torch.float32, device=dev)
        trace = torch.zeros(8 * GN_TRACE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        dbg = torch.zeros(48, dtype=torch.float32, device=dev) if dbg_level >= 0 else None
        _check(lib().svo_sparse_align(
            self._h, _imgs(prev_pyr, 8), _imgs(cur_pyr, 8), _ptr(kps2d), _ptr(kps3d), _ptr(flags),
            n, C.byref(cam), _ptr(pose_guess), _ptr(pose_out), _ptr(cost), _ptr(trace), _ptr(dbg),
            dbg_level))
        return pose_out, cost, trace, dbg

    # -- A3 ---------------------------------------------------------------
    def project_keypoints(self, pose, kps3d, cam):
        """project_keypoints (src/lib/transform_keypoints.cpp:11-48): pose [6] and kps3d device tensors."""
        n = kps3d.shape[0]
        out = torch.zeros((n, 2), dtype=torch.float32, device=kps3d.device)
        _check(lib().svo_project_keypoints(self._h, _ptr(pose), _ptr(kps3d), n, C.byref(cam), _ptr(out)))
        return out

    # -- B2 ---------------------------------------------------------------
    def klt_track(self, prev_lk, cur_lk, prev_pts, cur_pts, win):
        """OpticalFlow::calculate_optical_flow. cur_pts is updated in place."""
        n = prev_pts.shape[0]
        dev = prev_pts.device
        status = torch.zeros(n, dtype=torch.uint8, device=dev)
        err = torch.zeros(n, dtype=torch.float32, device=dev)
        nl = min(len(prev_lk), len(cur_lk))
        _check(lib().svo_klt_track(self._h, _imgs(prev_lk), _imgs(cur_lk), nl, _ptr(prev_pts),
                                   _ptr(cur_pts), n, win, _ptr(status), _ptr(err)))
        return cur_pts, status, err

    # -- B1 + B3 ----------------------------------------------------------
    def reproj_gn(self, kps2d, kps3d, flags, cam, pose_in, tracked=None, err=None):
        dev = kps2d.device
        n = kps2d.shape[0]
        pose_out = torch.zeros(6, dtype=torch.float32, device=dev)
        cost = torch.zeros(1, dtype=torch.float32, device=dev)
        trace = torch.zeros(GN_TRACE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        _check(lib().svo_reproj_gn(self._h, _ptr(kps2d), _ptr(kps3d), _ptr(flags), n, C.byref(cam),
                                   _ptr(tracked), _ptr(err), _ptr(pose_in), _ptr(pose_out),
                                   _ptr(cost), _ptr(trace)))
        return pose_out, cost, trace

    # -- C1 ---------------------------------------------------------------
    def ssd_disparity(self, left, right, kps2d, win, search_x, search_y, clamp_half=1):
        n = kps2d.shape[0]
        out = torch.zeros(n, dtype=torch.float32, device=kps2d.device)
        a, b = _img(left), _img(right)
        _check(lib().svo_ssd_disparity(self._h, C.byref(a), C.byref(b), _ptr(kps2d), n, win,
                                       search_x, search_y, clamp_half, _ptr(out)))
        return out

    # -- C2 + D1 ----------------------------------------------------------
    def depth_filter_update(self, kps2d, kps3d, flags, cam, frame_pose, disparity, ref3d, ref2d,
                            kf_pose, outlier, inlier, kf_x, kf_p, do_outlier_check=1, do_update=1):
        n = kps2d.shape[0]
        _check(lib().svo_depth_filter_update(
            self._h, _ptr(kps2d), _ptr(kps3d), _ptr(flags), n, C.byref(cam), _ptr(frame_pose),
            _ptr(disparity), _ptr(ref3d), _ptr(ref2d), _ptr(kf_pose), _ptr(outlier), _ptr(inlier),
            _ptr(kf_x), _ptr(kf_p), do_outlier_check, do_update))


def trace_to_numpy(t):
    return t.cpu().numpy().view(GN_TRACE_DTYPE)
