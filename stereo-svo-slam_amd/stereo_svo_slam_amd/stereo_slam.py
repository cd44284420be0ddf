"""Host-side mirror of the reference's user API on top of the C ABI.

`StereoSlam` has the live surface of the reference's Python wrapper
(src/python/wrapper/slam_accelerator.pyx:50-91: ctor(CameraSettings),
new_image(left, right, time_stamp), get_frame(), get_keyframe(), get_keyframes())
and of the C++ class (src/include/stereo_slam.hpp:35-62: + get_trajectory,
update_pose).  `StereoSlamBatch` drives B independent sequences through the same
kernel launches (one svo_ctx).  No compute happens in Python.
"""
import ctypes as C

import numpy as np
import torch

from . import hip_lib
from .hip_lib import CameraSettings, SvoError, _check, lib

KP_INFO_DTYPE = np.dtype([
    ("score", "<f4"), ("level", "<i4"), ("type", "<i4"), ("keyframe_id", "<i4"),
    ("keypoint_index", "<i4"), ("color", "u1", (3,)), ("ignore_during_refinement", "u1"),
    ("ignore_completely", "u1"), ("ignore_temporary", "u1"), ("_pad", "u1", (2,)),
    ("outlier_count", "<i4"), ("inlier_count", "<i4"), ("kf_inv_depth", "<f4"),
    ("kf_variance", "<f4")], align=False)
assert KP_INFO_DTYPE.itemsize == 44


class GnTrace(C.Structure):
    _fields_ = [("level", C.c_int32), ("n_gradient", C.c_int32), ("n_cost", C.c_int32),
                ("n_accepted", C.c_int32), ("exit_small", C.c_int32),
                ("initial_cost", C.c_float), ("final_cost", C.c_float), ("pose", C.c_float * 6)]


class FrameStats(C.Structure):
    """svo_frame_stats (include/svo_hip.h)."""
    _fields_ = [("frame_id", C.c_int32), ("is_keyframe", C.c_int32), ("n_keypoints", C.c_int32),
                ("n_keyframes", C.c_int32), ("inside_count", C.c_int32), ("overflow", C.c_int32),
                ("pose_sia", C.c_float * 6), ("pose_refined", C.c_float * 6),
                ("sia_cost", C.c_float), ("reproj_cost", C.c_float), ("sia_ms", C.c_float),
                ("stage_ms", C.c_float * 8),
                ("sia_trace", GnTrace * 8), ("reproj_trace", GnTrace)]


class Totals(C.Structure):
    """svo_totals (include/svo_hip.h)."""
    _fields_ = [("frames", C.c_int64), ("keyframes", C.c_int64), ("keypoints", C.c_int64),
                ("gn_gradient_calls", C.c_int64), ("gn_cost_calls", C.c_int64),
                ("stage_ms", C.c_double * 8), ("wall_ms", C.c_double),
                ("launches", C.c_int64), ("n_groups", C.c_int32), ("image_sets", C.c_int32)]


class Frame:
    """Frame / KeyFrame (src/include/stereo_slam_types.hpp:117-131) without images."""

    def __init__(self, pose, kps2d, kps3d, info):
        self.pose = pose
        self.kps2d = kps2d
        self.kps3d = kps3d
        self.info = info


class StereoSlamBatch:
    def __init__(self, camera_settings, width, height, n_sequences=1, device=0):
        if isinstance(camera_settings, dict):
            camera_settings = CameraSettings.from_dict(camera_settings)
        if not torch.cuda.is_available():
            raise SvoError("no GPU visible: libsvo_hip has no CPU fallback")
        self.cam = camera_settings
        self.width, self.height, self.n = width, height, n_sequences
        self.device = torch.device("cuda", device)
        self._ctx = C.c_void_p()
        _check(lib().svo_ctx_create(C.byref(self.cam), width, height, n_sequences, device,
                                    C.byref(self._ctx)))

    def close(self):
        if getattr(self, "_ctx", None) and hip_lib._LIB is not None:
            hip_lib._LIB.svo_ctx_destroy(self._ctx)
        self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_fast_solver(self, on=True):
        """svo_ctx_set_fast_solver: off (default) = reference-order Gauss-Newton (bit-exact traces);
        on = tree sums + LDL^T solve."""
        self._fast = bool(on)
        if self._ctx:
            _check(lib().svo_ctx_set_fast_solver(self._ctx, int(on)))

    def set_exact_pinv(self, on=True):
        self.set_fast_solver(not on)

    def enable_timing(self, on=True):
        self._timing = bool(on)
        if self._ctx:                      # (StereoSlam creates its ctx with the first image)
            _check(lib().svo_ctx_enable_timing(self._ctx, int(on)))

    def new_images(self, lefts, rights, time_stamps):
        """lefts/rights: per sequence a uint8 [H, W] numpy array (host) or torch CUDA tensor; None for
        a sequence that has no frame at this step (it sits the step out: sequences of a ctx may have
        different lengths)."""
        assert len(lefts) == self.n and len(rights) == self.n
        on_dev = isinstance(next(x for x in lefts if x is not None), torch.Tensor)
        ptrs_l = (C.c_void_p * self.n)()
        ptrs_r = (C.c_void_p * self.n)()
        keep = []
        stride = None
        for s in range(self.n):
            if lefts[s] is None:
                continue
            for arr, dst in ((lefts[s], ptrs_l), (rights[s], ptrs_r)):
                if on_dev:
                    assert arr.is_cuda and arr.dtype == torch.uint8 and arr.stride(1) == 1
                    st, p = arr.stride(0), arr.data_ptr()
                else:
                    arr = np.ascontiguousarray(arr, dtype=np.uint8)
                    st, p = arr.strides[0], arr.ctypes.data
                assert tuple(arr.shape) == (self.height, self.width)
                assert stride is None or stride == st
                stride = st
                keep.append(arr)
                dst[s] = p
        if on_dev:
            torch.cuda.current_stream(self.device).synchronize()
        ts = (C.c_float * self.n)(*[float(t) for t in time_stamps])
        _check(lib().svo_new_images(self._ctx, ptrs_l, ptrs_r, stride, ts, 1 if on_dev else 0))

    def pack_images(self, lefts, rights, time_stamps, borrow=False):
        """Pre-build the argument arrays of one step (keeps Python out of a timed loop); pass the
        result to new_images_packed / submit_packed. The frames are torch uint8 tensors, all on
        the GPU (SVO_MEM_DEVICE; with borrow=True SVO_MEM_DEVICE_BORROW: used in place, the caller
        keeps them alive and unchanged) or all in host memory (SVO_MEM_HOST; pinned for full PCIe rate)."""
        ptrs_l = (C.c_void_p * self.n)()
        ptrs_r = (C.c_void_p * self.n)()
        some = next(x for x in lefts if x is not None)
        stride = some.stride(0)
        on_dev = some.is_cuda
        for s in range(self.n):
            if lefts[s] is None:                            # the sequence sits this step out
                continue
            for arr, dst in ((lefts[s], ptrs_l), (rights[s], ptrs_r)):
                assert arr.is_cuda == on_dev and arr.dtype == torch.uint8 and arr.stride(1) == 1
                assert tuple(arr.shape) == (self.height, self.width) and arr.stride(0) == stride
                dst[s] = arr.data_ptr()
        ts = (C.c_float * self.n)(*[float(t) for t in time_stamps])
        return ptrs_l, ptrs_r, stride, ts, (lefts, rights), (2 if borrow else 1) if on_dev else 0

    def new_images_packed(self, packed):
        _check(lib().svo_new_images(self._ctx, packed[0], packed[1], packed[2], packed[3], packed[5]))

    def submit_packed(self, packed):
        """Pipelined form (svo_submit_images): queues the frame set on every sequence group and
        returns; `packed` (and its device images) must stay alive until wait()."""
        _check(lib().svo_submit_images(self._ctx, packed[0], packed[1], packed[2], packed[3], packed[5]))

    def wait(self):
        _check(lib().svo_wait(self._ctx))

    def groups(self):
        n = C.c_int(0)
        _check(lib().svo_ctx_get_groups(self._ctx, C.byref(n)))
        return n.value

    def totals(self):
        t = Totals()
        _check(lib().svo_get_totals(self._ctx, C.byref(t)))
        return t

    def pose(self, seq=0):
        p = np.zeros(6, np.float32)
        _check(lib().svo_get_pose(self._ctx, seq, p.ctypes.data_as(C.c_void_p)))
        return p

    def stats(self, seq=0):
        st = FrameStats()
        _check(lib().svo_get_frame_stats(self._ctx, seq, C.byref(st)))
        return st

    def get_frame(self, seq=0):
        n = C.c_int(0)
        _check(lib().svo_get_frame_keypoints(self._ctx, seq, None, None, None, 0, C.byref(n)))
        k2 = np.zeros((n.value, 2), np.float32)
        k3 = np.zeros((n.value, 3), np.float32)
        info = np.zeros(n.value, KP_INFO_DTYPE)
        _check(lib().svo_get_frame_keypoints(self._ctx, seq, k2.ctypes.data_as(C.c_void_p),
                                             k3.ctypes.data_as(C.c_void_p),
                                             info.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return Frame(self.pose(seq), k2, k3, info)

    def num_keyframes(self, seq=0):
        n = C.c_int(0)
        _check(lib().svo_get_keyframe_count(self._ctx, seq, C.byref(n)))
        return n.value

    def get_keyframe(self, kid=None, seq=0):
        if kid is None:
            kid = self.num_keyframes(seq) - 1
        n = C.c_int(0)
        pose = np.zeros(6, np.float32)
        _check(lib().svo_get_keyframe(self._ctx, seq, kid, None, None, None,
                                      pose.ctypes.data_as(C.c_void_p), 0, C.byref(n)))
        k2 = np.zeros((n.value, 2), np.float32)
        k3 = np.zeros((n.value, 3), np.float32)
        info = np.zeros(n.value, KP_INFO_DTYPE)
        _check(lib().svo_get_keyframe(self._ctx, seq, kid, k2.ctypes.data_as(C.c_void_p),
                                      k3.ctypes.data_as(C.c_void_p), info.ctypes.data_as(C.c_void_p),
                                      pose.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return Frame(pose, k2, k3, info)

    def get_keyframes(self, seq=0):
        return [self.get_keyframe(i, seq) for i in range(self.num_keyframes(seq))]

    def get_trajectory(self, seq=0):
        n = C.c_int(0)
        _check(lib().svo_get_trajectory(self._ctx, seq, None, 0, C.byref(n)))
        out = np.zeros((n.value, 6), np.float32)
        _check(lib().svo_get_trajectory(self._ctx, seq, out.ctypes.data_as(C.c_void_p), n.value,
                                        C.byref(n)))
        return out

    def update_pose(self, pose, speed, pose_variance, speed_variance, dt, seq=0):
        out = np.zeros(6, np.float32)
        arrs = [np.ascontiguousarray(a, np.float32) for a in (pose, speed, pose_variance, speed_variance)]
        _check(lib().svo_update_pose(self._ctx, seq, *[a.ctypes.data_as(C.c_void_p) for a in arrs],
                                     C.c_double(dt), out.ctypes.data_as(C.c_void_p)))
        return out


class StereoSlam(StereoSlamBatch):
    """One sequence: the reference's StereoSlam."""

    def __init__(self, camera_settings, width=None, height=None, device=0):
        self._pending = (camera_settings, device)
        self._ctx = None
        if width is not None:
            super().__init__(camera_settings, width, height, 1, device)

    def new_image(self, left, right, time_stamp):
        if self._ctx is None:   # the reference learns the image size from the first frame
            cam, device = self._pending
            h, w = left.shape
            super().__init__(cam, w, h, 1, device)
            if getattr(self, "_fast", False):
                self.set_fast_solver(True)
            if getattr(self, "_timing", False):
                self.enable_timing(True)
        self.new_images([left], [right], [time_stamp])
