"""The C++ facade classes with the reference's names (hostcpp/: StereoSlam, PoseEstimator,
project_keypoints, PoseRefiner + OpticalFlow, DepthFilter; reference:
src/include/pose_estimator.hpp:19-27, pose_refinement.hpp:22-32, optical_flow.hpp:26-30,
depth_filter.hpp:14-20) built with g++ against libsvo_hip.so and run on the GPU; their results
must equal, bit for bit, what the ctypes binding gets from the same C entry points."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from stereo_svo_slam_amd import hip_lib, synth
from stereo_svo_slam_amd.stereo_slam import StereoSlam
import util

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stereo-svo-slam_amd", "csrc")


def _build(tmp_path, name):
    exe = str(tmp_path / name)
    gxx = shutil.which("g++")
    assert gxx
    subprocess.check_call([gxx, "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", name + ".cpp"),
                           "-o", exe, "-L" + CSRC, "-lsvo_hip", "-Wl,-rpath," + CSRC,
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_facade_smoke_runs_on_the_gpu(tmp_path):
    exe = _build(tmp_path, "facade_smoke")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "keypoints" in r.stdout, (r.returncode, r.stdout, r.stderr)


@pytest.mark.parametrize("exact", [False, True])
def test_facade_stage_classes_equal_the_ctypes_path(tmp_path, exact):
    cfg, L, R, poses, ts = synth.make_sequence("tiny", 2, 0, device="cpu")
    L = [x.numpy() for x in L]
    R = [x.numpy() for x in R]
    w, h = cfg["width"], cfg["height"]
    cam = hip_lib.CameraSettings.from_dict(cfg)
    for name, img in (("l0", L[0]), ("r0", R[0]), ("l1", L[1]), ("r1", R[1])):
        img.tofile(str(tmp_path / (name + ".raw")))
    with open(tmp_path / "cam.bin", "wb") as f:
        f.write(bytes(cam))
    exe = _build(tmp_path, "facade_stages")
    r = subprocess.run([exe, str(tmp_path), str(w), str(h)] + ([] if exact else ["fast"]),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype=np.uint8)
    n = int(raw[:4].view(np.int32)[0])
    fl = raw[4:].view(np.float32)
    o = 0

    def take(k):
        nonlocal o
        v = fl[o:o + k]
        o += k
        return v

    pose_sia, sia_cost = take(6), take(1)[0]
    pose_ref, ref_cost = take(6), take(1)[0]
    projected = take(2 * n).reshape(n, 2)
    merged = take(2 * n).reshape(n, 2)
    updated = take(3 * n).reshape(n, 3)
    rec = take(5 * n).reshape(n, 5)

    # the same steps through the ctypes binding
    slam = StereoSlam(cfg)
    slam.new_image(L[0], R[0], 0.0)
    f0 = slam.get_frame()
    assert len(f0.kps2d) == n
    H = hip_lib.Handle(0, 4096)
    H.set_exact_pinv(exact)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    nl = cfg["max_pyramid_levels"]
    win = cfg["window_size_opt_flow"]
    p0, p1 = H.build_pyramid(dev(L[0]), nl), H.build_pyramid(dev(L[1]), nl)
    k2, k3 = dev(f0.kps2d), dev(f0.kps3d)
    flags = dev(util.flags_of(f0.info))
    g_pose, g_cost, _, _ = H.sparse_align(p0, p1, k2, k3, flags, cam, dev(np.zeros(6, np.float32)))
    assert np.array_equal(g_pose.cpu().numpy(), pose_sia) and float(g_cost.cpu()) == sia_cost
    g_proj = H.project_keypoints(g_pose, k3, cam)
    assert np.array_equal(g_proj.cpu().numpy(), projected)
    lk0, lk1 = H.build_lk_pyramid(dev(L[0]), win), H.build_lk_pyramid(dev(L[1]), win)
    cur = g_proj.clone()
    _, st, err = H.klt_track(lk0, lk1, k2, cur, win)       # one keyframe: every point tracks against it
    g_k2 = g_proj.clone()
    g_fl = flags.clone()
    g_ref, g_rcost, _ = H.reproj_gn(g_k2, k3, g_fl, cam, g_pose, cur, err)
    assert np.array_equal(g_ref.cpu().numpy(), pose_ref) and float(g_rcost.cpu()) == ref_cost
    assert np.array_equal(g_k2.cpu().numpy(), merged)
    disp = H.ssd_disparity(dev(L[1]), dev(R[1]), g_k2, cfg["window_size_depth_calculator"], cfg["search_x"],
                           cfg["search_y"], 1)
    g3 = k3.clone()
    outl = dev(f0.info["outlier_count"].astype(np.int32))
    inl = dev(f0.info["inlier_count"].astype(np.int32))
    kx, kP = dev(f0.info["kf_inv_depth"].copy()), dev(f0.info["kf_variance"].copy())
    kf_pose = dev(np.zeros((n, 6), np.float32))
    H.depth_filter_update(g_k2, g3, g_fl, cam, g_ref, disp, k3, k2, kf_pose, outl, inl, kx, kP)
    assert np.array_equal(g3.cpu().numpy(), updated)
    assert np.array_equal(g_fl.cpu().numpy().astype(np.float32), rec[:, 0])
    assert np.array_equal(outl.cpu().numpy().astype(np.float32), rec[:, 1])
    assert np.array_equal(inl.cpu().numpy().astype(np.float32), rec[:, 2])
    assert np.array_equal(kx.cpu().numpy(), rec[:, 3]) and np.array_equal(kP.cpu().numpy(), rec[:, 4])
    assert np.linalg.norm(pose_ref[:3] - poses[1][:3]) < 0.05        # and it is a sensible pose
    H.close()
