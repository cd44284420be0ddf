"""Seeded synthetic stereo sequences (no dataset ships with the reference:
the Blender .mkv files are stripped and EuRoC is not redistributable).

A closed room of textured planes plus a few free-standing panels is ray-cast
into a rectified pinhole stereo pair with the reference's conventions:

* pose = camera in world, point_cam = R(-r) (P - t)   (src/lib/transform_keypoints.cpp:32-45)
* the library's `left` image is the camera at the pose; `right` is displaced
  by -baseline/fx along the camera x axis, so a point at column u in `left`
  appears at u + baseline/z in `right` (the disparity search of
  src/lib/depth_filter.cpp:293-302 runs toward +x).

Pure torch, runs on CPU (tests) and on the GPU (bench); used for inputs only.
"""
import math

import numpy as np
import torch

# camera presets: BASELINE.json configs mapped onto the reference's YAMLs
# (src/app/EuRoC.yaml:35-43, src/app/Blender.yaml:42-59, src/app/Econ.yaml:9-48)
CONFIGS = {
    # C2: EuRoC MH_02 class, 752x480, 6/2 levels ("4-level"), ~130-200 patches
    "euroc": dict(width=752, height=480, fx=435.2046959714599, fy=435.2046959714599,
                  cx=367.4517211914062, cy=252.2008514404297, baseline=47.90639384423901,
                  k1=0.0, k2=0.0, k3=0.0, p1=0.0, p2=0.0, grid_width=54, grid_height=48,
                  search_x=60, search_y=6, window_size_pose_estimator=4,
                  window_size_opt_flow=31, window_size_depth_calculator=31,
                  max_pyramid_levels=6, min_pyramid_level_pose_estimation=2),
    # C1: Blender classroom, 752x480, 5/2 levels ("3-level")
    "blender": dict(width=752, height=480, fx=470.0, fy=470.0, cx=376.0, cy=240.0, baseline=28.2,
                    k1=0.0, k2=0.0, k3=0.0, p1=0.0, p2=0.0, grid_width=75, grid_height=48,
                    search_x=50, search_y=6, window_size_pose_estimator=4,
                    window_size_opt_flow=31, window_size_depth_calculator=31,
                    max_pyramid_levels=5, min_pyramid_level_pose_estimation=2),
    # C3: synthetic 1920x1080, 2000 patches, 7/2 levels ("5-level")
    "hd": dict(width=1920, height=1080, fx=1200.0, fy=1200.0, cx=960.0, cy=540.0, baseline=72.0,
               k1=0.0, k2=0.0, k3=0.0, p1=0.0, p2=0.0, grid_width=43, grid_height=24,
               search_x=60, search_y=6, window_size_pose_estimator=4,
               window_size_opt_flow=31, window_size_depth_calculator=31,
               max_pyramid_levels=7, min_pyramid_level_pose_estimation=2),
    # C5: Econ Tara, distortion != 0 (applied in projection only), windows 35
    "econ": dict(width=752, height=480, fx=743.8041254687444, fy=743.8041254687444,
                 cx=365.86266803741455, cy=238.70182609558105, baseline=45.1932,
                 k1=0.12598132, k2=-0.22447148, k3=0.09229389, p1=0.00074527, p2=0.00802387,
                 grid_width=40, grid_height=50, search_x=60, search_y=6,
                 window_size_pose_estimator=4, window_size_opt_flow=35,
                 window_size_depth_calculator=35, max_pyramid_levels=5,
                 min_pyramid_level_pose_estimation=2),
    # small case for fast CPU tests
    "tiny": dict(width=320, height=240, fx=200.0, fy=200.0, cx=160.0, cy=120.0, baseline=20.0,
                 k1=0.0, k2=0.0, k3=0.0, p1=0.0, p2=0.0, grid_width=40, grid_height=40,
                 search_x=30, search_y=4, window_size_pose_estimator=4,
                 window_size_opt_flow=21, window_size_depth_calculator=21,
                 max_pyramid_levels=4, min_pyramid_level_pose_estimation=1),
}

CAMERA_FIELDS = ("baseline", "fx", "fy", "cx", "cy", "k1", "k2", "k3", "p1", "p2",
                 "grid_height", "grid_width", "search_x", "search_y",
                 "window_size_pose_estimator", "window_size_opt_flow",
                 "window_size_depth_calculator", "max_pyramid_levels",
                 "min_pyramid_level_pose_estimation")


def rodrigues(r):
    """float64 3x3 rotation of the axis-angle vector r (numpy)."""
    r = np.asarray(r, np.float64)
    th = np.linalg.norm(r)
    if th < 1e-15:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return math.cos(th) * np.eye(3) + (1 - math.cos(th)) * np.outer(k, k) + math.sin(th) * K


def trajectory(n_frames, seed=0, scale=1.0):
    """Smooth 6-DoF camera path, <= ~2 cm and <= ~0.2 deg per frame; frame 0 at the origin."""
    rng = np.random.RandomState(1000 + seed)
    ph = rng.uniform(0, 2 * math.pi, 6)
    k = np.arange(n_frames, dtype=np.float64)[:, None]
    amp = np.array([0.25, 0.10, 0.20, 0.03, 0.05, 0.02]) * scale
    om = np.array([0.05, 0.08, 0.04, 0.06, 0.045, 0.07])
    p = amp * (np.sin(om * k + ph) - np.sin(ph))
    return p.astype(np.float32)


def loop_trajectory(n_frames, seed=0, scale=1.0):
    """Closed 6-DoF camera path of n_frames poses: every component is a sum of sinusoids whose periods
    divide n_frames, so pose[n_frames] == pose[0] and the frames can be played round and round, from any
    starting frame, always forward. Sideways / forward drift of about +-0.6 m, yaw of about +-0.5 rad
    and smaller pitch / roll terms: at 192 frames per lap <= ~4 cm and <= ~1.1 deg per frame (a brisk
    hand-held / MAV motion), enough image motion that the reference's keyframe rule
    (src/lib/keyframe_manager.cpp:47-74: fewer than 66 % of the grid cells hold a visible point) fires
    every few tens of frames, as it does on the reference's own sequences."""
    rng = np.random.RandomState(3000 + seed)
    ph = rng.uniform(0, 2 * math.pi, 12)
    a = 2 * math.pi * np.arange(n_frames, dtype=np.float64) / n_frames
    x = 0.55 * np.sin(a + ph[0]) + 0.10 * np.sin(3 * a + ph[1])
    y = 0.10 * np.sin(2 * a + ph[2]) + 0.04 * np.sin(5 * a + ph[3])
    z = 0.45 * np.sin(a + ph[0] + 1.3) + 0.10 * np.sin(2 * a + ph[4])
    rx = 0.10 * np.sin(3 * a + ph[5]) + 0.03 * np.sin(7 * a + ph[6])
    ry = 0.42 * np.sin(2 * a + ph[7]) + 0.10 * np.sin(3 * a + ph[8])
    rz = 0.05 * np.sin(4 * a + ph[9]) + 0.02 * np.sin(6 * a + ph[10])
    p = np.stack([x, y, z, rx, ry, rz], 1) * scale
    return p.astype(np.float32)


def _texture(rng, size=1024):
    t = np.full((size, size), 128.0, np.float32)
    for s, a in ((128, 28.0), (64, 26.0), (32, 24.0), (16, 20.0), (8, 14.0), (4, 8.0)):
        # (np.kron with a block of ones, written as two repeats: same values, ~50x faster)
        t += rng.uniform(-a, a, (size // s, size // s)).astype(np.float32).repeat(s, 0).repeat(s, 1)
    t = 0.25 * (t + np.roll(t, 1, 0) + np.roll(t, 1, 1) + np.roll(np.roll(t, 1, 0), 1, 1))
    return np.clip(t, 16, 240)


class Scene:
    """Planes: (p0, u, v, half_u, half_v) with u,v orthonormal in-plane axes;
    half extents <= 0 mean unbounded."""

    def __init__(self, seed=0, device="cpu", texels_per_meter=170.0):
        rng = np.random.RandomState(20241004 + seed)
        self.device = torch.device(device)
        self.tpm = texels_per_meter
        ex, ey, ez = np.eye(3)
        planes = [
            ((0, 0, 6.5), ex, ey, 0, 0),        # back wall
            ((0, 1.6, 0), ex, ez, 0, 0),        # floor (y is down)
            ((0, -1.7, 0), ex, ez, 0, 0),       # ceiling
            ((-3.2, 0, 0), ez, ey, 0, 0),       # left wall
            ((3.4, 0, 0), ez, ey, 0, 0),        # right wall
            ((0, 0, -3.0), ex, ey, 0, 0),       # wall behind the camera
        ]
        for _ in range(5):                       # free-standing panels
            c = (rng.uniform(-2.0, 2.0), rng.uniform(-0.8, 0.9), rng.uniform(2.2, 4.8))
            yaw = rng.uniform(-0.5, 0.5)
            u = np.array([math.cos(yaw), 0, math.sin(yaw)])
            planes.append((c, u, ey, rng.uniform(0.35, 0.8), rng.uniform(0.3, 0.7)))
        self.planes = []
        for p0, u, v, hu, hv in planes:
            tex = torch.from_numpy(_texture(rng)).to(self.device)
            self.planes.append(dict(
                p0=torch.tensor(p0, dtype=torch.float64, device=self.device),
                u=torch.tensor(np.asarray(u, np.float64), device=self.device),
                v=torch.tensor(np.asarray(v, np.float64), device=self.device),
                hu=float(hu), hv=float(hv), tex=tex,
                off=(float(rng.uniform(0, 1024)), float(rng.uniform(0, 1024)))))

    def render_batch(self, cfg, poses, right=False):
        """float32 [K, height, width] noise-free images of the camera at poses [K, 6].
        Per-pixel math is float32 and purely elementwise (no BLAS calls)."""
        dev = self.device
        w, h = cfg["width"], cfg["height"]
        poses = np.asarray(poses, np.float64).reshape(-1, 6)
        K = poses.shape[0]
        Rn = np.stack([rodrigues(p[3:6]) for p in poses])                  # [K,3,3] float64
        on = poses[:, 0:3].copy()
        if right:
            b = cfg["baseline"] / cfg["fx"]
            on = on + Rn @ np.array([-b, 0.0, 0.0])
        f32 = torch.float32
        R = torch.tensor(Rn, dtype=f32, device=dev)
        o = torch.tensor(on, dtype=f32, device=dev)
        xs = ((torch.arange(w, dtype=torch.float64, device=dev) - cfg["cx"]) / cfg["fx"]).to(f32)
        ys = ((torch.arange(h, dtype=torch.float64, device=dev) - cfg["cy"]) / cfg["fy"]).to(f32)
        X = xs[None, None, :]
        Y = ys[None, :, None]
        # world ray directions d = R (x, y, 1), one [K,h,w] tensor per component
        d = [R[:, i, 0, None, None] * X + R[:, i, 1, None, None] * Y + R[:, i, 2, None, None]
             for i in range(3)]
        best_s = torch.full((K, h, w), float("inf"), dtype=f32, device=dev)
        img = torch.zeros(K, h, w, dtype=f32, device=dev)
        for pl in self.planes:
            u, v, p0 = (pl[k].to(f32) for k in ("u", "v", "p0"))
            n = torch.linalg.cross(u, v)
            dn = d[0] * n[0] + d[1] * n[1] + d[2] * n[2]
            num = ((p0[None, :] - o) * n[None, :]).sum(1)                  # [K]
            s = num[:, None, None] / dn
            q = [o[:, i, None, None] + s * d[i] - p0[i] for i in range(3)]
            tu = q[0] * u[0] + q[1] * u[1] + q[2] * u[2]
            tv = q[0] * v[0] + q[1] * v[1] + q[2] * v[2]
            del q
            ok = (s > 1e-3) & (s < best_s) & torch.isfinite(s)
            if pl["hu"] > 0:
                ok &= (tu.abs() < pl["hu"]) & (tv.abs() < pl["hv"])
            fu = tu * self.tpm + pl["off"][0]
            fv = tv * self.tpm + pl["off"][1]
            del tu, tv
            iu, iv = torch.floor(fu), torch.floor(fv)
            au, av = fu - iu, fv - iv
            size = pl["tex"].shape[0]
            iu0 = torch.remainder(iu, size).to(torch.int32)
            iv0 = torch.remainder(iv, size).to(torch.int32)
            del fu, fv, iu, iv
            iu0 = torch.where(ok, iu0, 0)
            iv0 = torch.where(ok, iv0, 0)
            iu1 = torch.remainder(iu0 + 1, size)
            iv1 = torch.remainder(iv0 + 1, size)
            t = pl["tex"].reshape(-1)
            i00 = (iv0 * size + iu0).long(); i01 = (iv0 * size + iu1).long()
            i10 = (iv1 * size + iu0).long(); i11 = (iv1 * size + iu1).long()
            val = (t[i00] * (1 - au) * (1 - av) + t[i01] * au * (1 - av) +
                   t[i10] * (1 - au) * av + t[i11] * au * av)
            img = torch.where(ok, val, img)
            best_s = torch.where(ok, s, best_s)
        return img

    def render(self, cfg, pose, right=False, noise_sigma=1.0, noise_seed=None):
        """uint8 (height, width) image of the camera at `pose` (6 floats)."""
        dev = self.device
        w, h = cfg["width"], cfg["height"]
        img = self.render_batch(cfg, np.asarray(pose, np.float64)[None], right)[0]
        if noise_sigma > 0:
            if noise_seed is None:
                img = img + noise_sigma * torch.randn(h, w, device=dev)
            else:
                g = np.random.RandomState(noise_seed).standard_normal((h, w)).astype(np.float32)
                img = img + noise_sigma * torch.from_numpy(g).to(dev)
        return img.round().clamp(0, 255).to(torch.uint8)


# ---------------------------------------------------------------- fused GPU renderer
class _SynthPlane(__import__("ctypes").Structure):
    _fields_ = [("p0", __import__("ctypes").c_float * 3), ("u", __import__("ctypes").c_float * 3),
                ("v", __import__("ctypes").c_float * 3), ("hu", __import__("ctypes").c_float),
                ("hv", __import__("ctypes").c_float), ("off_u", __import__("ctypes").c_float),
                ("off_v", __import__("ctypes").c_float)]


class _SynthParams(__import__("ctypes").Structure):
    _C = __import__("ctypes")
    _fields_ = [("planes", _SynthPlane * 16), ("n_planes", _C.c_int), ("tex_size", _C.c_int),
                ("tpm", _C.c_float), ("fx", _C.c_float), ("fy", _C.c_float), ("cx", _C.c_float),
                ("cy", _C.c_float), ("w", _C.c_int), ("h", _C.c_int), ("noise_sigma", _C.c_float),
                ("seed", _C.c_uint32)]


_SYNTH_LIB = None


def _synth_lib():
    """csrc/libsvo_synth.so (synth_render.hip): one kernel per batch of frames instead of ~450
    elementwise torch launches. Input generation only."""
    global _SYNTH_LIB
    if _SYNTH_LIB is None:
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "csrc", "libsvo_synth.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run __graft_entry__.build()")
        _SYNTH_LIB = C.CDLL(path)
    return _SYNTH_LIB


def render_frames_gpu(scene, cfg, poses, right=False, noise_sigma=1.0, noise_seeds=None, out=None):
    """uint8 [K, height, width] CUDA tensor: `scene` seen from `poses` [K, 6] (the library's left
    camera, or the right one), rendered by one launch of csrc/synth_render.hip."""
    import ctypes as C
    dev = scene.device
    assert dev.type == "cuda", "render_frames_gpu needs a scene on the GPU"
    w, h = cfg["width"], cfg["height"]
    poses = np.asarray(poses, np.float64).reshape(-1, 6)
    K = poses.shape[0]
    Rn = np.stack([rodrigues(p[3:6]) for p in poses])
    on = poses[:, 0:3].copy()
    if right:
        on = on + Rn @ np.array([-cfg["baseline"] / cfg["fx"], 0.0, 0.0])
    host = np.concatenate([Rn.reshape(K, 9), on], 1).astype(np.float32)
    if not hasattr(scene, "_tex_stack"):
        scene._tex_stack = torch.stack([pl["tex"] for pl in scene.planes]).contiguous()
        par = _SynthParams()                      # plane table: built once per scene (reads device tensors)
        par.n_planes = len(scene.planes)
        par.tex_size = scene.planes[0]["tex"].shape[0]
        par.tpm = scene.tpm
        for i, pl in enumerate(scene.planes):
            sp = par.planes[i]
            p0, u, v = (pl[k].cpu().numpy() for k in ("p0", "u", "v"))
            for j in range(3):
                sp.p0[j] = float(p0[j]); sp.u[j] = float(u[j]); sp.v[j] = float(v[j])
            sp.hu, sp.hv = pl["hu"], pl["hv"]
            sp.off_u, sp.off_v = pl["off"]
        scene._synth_par = par
    par = scene._synth_par
    par.fx, par.fy, par.cx, par.cy = cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"]
    par.w, par.h = w, h
    par.noise_sigma = noise_sigma
    par.seed = 1 if right else 0
    d_poses = torch.from_numpy(host).to(dev)
    if noise_seeds is None:
        noise_seeds = np.arange(K)
    d_seeds = torch.from_numpy(np.asarray(noise_seeds, np.int64).astype(np.uint32).view(np.int32).copy()).to(dev)
    if out is None:
        out = torch.empty((K, h, w), dtype=torch.uint8, device=dev)
    rc = _synth_lib().svo_synth_render(C.byref(par), C.c_void_p(scene._tex_stack.data_ptr()),
                                       C.c_void_p(d_poses.data_ptr()), C.c_void_p(d_seeds.data_ptr()), K,
                                       C.c_void_p(out.data_ptr()),
                                       C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"svo_synth_render failed ({rc})")
    return out


def make_sequence_gpu(config="euroc", n_frames=10, seed=0, device="cuda", noise_sigma=1.0, motion_scale=1.0):
    """make_sequence with the fused renderer (same scene and path, its own noise): returns
    (cfg, lefts [n,H,W] uint8 CUDA tensor, rights, poses, timestamps)."""
    cfg = dict(CONFIGS[config])
    scene = Scene(seed, device)
    poses = trajectory(n_frames, seed, motion_scale)
    seeds = 7919 * (seed + 1) + 2 * np.arange(n_frames)
    lefts = render_frames_gpu(scene, cfg, poses, False, noise_sigma, seeds)
    rights = render_frames_gpu(scene, cfg, poses, True, noise_sigma, seeds + 1)
    return cfg, lefts, rights, poses, np.arange(n_frames, dtype=np.float32) / 20.0


def make_sequence(config="euroc", n_frames=10, seed=0, device="cpu", noise_sigma=1.0,
                  motion_scale=1.0):
    """Returns (cfg, lefts, rights, poses, timestamps); lefts/rights are lists
    of uint8 torch tensors on `device`; poses float32 [n,6] ground truth."""
    cfg = dict(CONFIGS[config])
    scene = Scene(seed, device)
    poses = trajectory(n_frames, seed, motion_scale)
    lefts, rights = [], []
    for k in range(n_frames):
        ns = None if noise_sigma <= 0 else 7919 * (seed + 1) + 2 * k
        lefts.append(scene.render(cfg, poses[k], False, noise_sigma, ns))
        rights.append(scene.render(cfg, poses[k], True, noise_sigma, None if ns is None else ns + 1))
    ts = np.arange(n_frames, dtype=np.float32) / 20.0
    return cfg, lefts, rights, poses, ts
