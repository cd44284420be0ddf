"""GPU parity: every stage of the hot path, called through the C ABI
(libsvo_hip.so), against the CPU oracle on the same seeded inputs.

Integer / index results must be bit-exact; float results use the tolerances
written next to each assert (pose <= 1e-4 m / rad).
"""
import numpy as np
import pytest
import torch

import oracle_py as O
from stereo_svo_slam_amd import hip_lib, synth
import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    h = hip_lib.Handle(0, max_keypoints=4096)
    yield h
    h.close()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def cam_of(cfg):
    return hip_lib.CameraSettings.from_dict(cfg)


# ------------------------------------------------------------------ P1 / P2
# Two kernels build the pyramids: the row-streaming one (width a multiple of 8, even height, aligned
# rows: every configuration of the reference; row blocks of 32 up to six levels, of 64 with seven or with
# SVO_PYR_KERNEL=64) and the tile kernel (anything else, or SVO_PYR_KERNEL=tile).
STREAM_SHAPES = [((480, 752), 6), ((240, 320), 4), ((1080, 1920), 7), ((64, 64), 4), ((48, 456), 4),
                 ((130, 904), 5), ((16, 16), 3), ((66, 3584), 6)]


@pytest.mark.parametrize("kernel", ["stream", "64", "tile"])
@pytest.mark.parametrize("shape,levels", STREAM_SHAPES + [((131, 203), 5)])
def test_halfsample_pyramid_bit_exact(H, shape, levels, kernel, monkeypatch):
    monkeypatch.setenv("SVO_PYR_KERNEL", kernel)
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, shape).astype(np.uint8)
    ref = O.build_pyramid(img, levels)
    got = H.build_pyramid(dev(img), levels)
    for l in range(levels):
        assert np.array_equal(got[l].cpu().numpy(), ref[l]), f"level {l}"


def test_halfsample_pyramid_real_image(H):
    left, _ = util.real_pair()
    ref = O.build_pyramid(left, 6)
    got = H.build_pyramid(dev(left), 6)
    for l in range(6):
        assert np.array_equal(got[l].cpu().numpy(), ref[l])


@pytest.mark.parametrize("kernel", ["stream", "64", "tile"])
@pytest.mark.parametrize("shape,win", [((480, 752), 31), ((240, 320), 21), ((97, 131), 21),
                                       ((1080, 1920), 31), ((64, 72), 9), ((130, 904), 31), ((66, 456), 9),
                                       ((34, 3584), 5), ((482, 752), 31), ((476, 752), 31)])
def test_lk_pyramid_bit_exact(H, shape, win, kernel, monkeypatch):
    monkeypatch.setenv("SVO_PYR_KERNEL", kernel)
    rng = np.random.RandomState(2)
    img = rng.randint(0, 256, shape).astype(np.uint8)
    ref = O.build_lk_pyramid(img, win)
    got = H.build_lk_pyramid(dev(img), win)
    assert len(got) == len(ref)
    for l in range(len(ref)):
        assert np.array_equal(got[l].cpu().numpy(), ref[l]), f"level {l}"


# ----------------------------------------------------------------------- C1
def _ssd_case(H, left, right, kps, win, sx, sy, clamp):
    ref = O.ssd_disparity(left, right, kps, win, sx, sy, clamp)
    got = H.ssd_disparity(dev(left), dev(right), dev(kps), win, sx, sy, clamp).cpu().numpy()
    assert np.array_equal(got, ref), (got[got != ref][:8], ref[got != ref][:8])


def test_ssd_disparity_real_pair_bit_exact(H):
    left, right = util.real_pair()
    rng = np.random.RandomState(3)
    kps = np.stack([rng.uniform(0, 752, 300), rng.uniform(0, 480, 300)], 1).astype(np.float32)
    _ssd_case(H, left, right, kps, 31, 60, 6, 1)
    _ssd_case(H, left, right, kps, 35, 60, 6, 1)
    _ssd_case(H, left, right, kps, 33, 64, 8, 1)      # 17 x 65 match map: every block of the MFMA tiling
    _ssd_case(H, left, right, kps, 5, 3, 1, 0)


def test_ssd_disparity_borders_and_ties(H):
    # constant images: every offset ties -> tie-averaged column; keypoints on and off the border
    left = np.full((120, 160), 77, np.uint8)
    right = np.full((120, 160), 77, np.uint8)
    kps = np.array([[0, 0], [159.9, 119.9], [5, 60], [80, 3], [158, 60], [80, 118], [-20, 10],
                    [200, 50], [80, 60]], np.float32)
    _ssd_case(H, left, right, kps, 21, 30, 4, 1)
    rng = np.random.RandomState(4)
    left = rng.randint(0, 256, (120, 160)).astype(np.uint8)
    right = np.roll(left, 7, axis=1)
    kin = kps[(kps[:, 0] >= 0) & (kps[:, 0] < 160)]
    _ssd_case(H, left, right, kps, 21, 30, 4, 1)
    _ssd_case(H, left, right, kin, 21, 30, 4, 0)


def test_ssd_disparity_synthetic_scene(H):
    sc = util.scenario("euroc", 2, 0, 1)
    cfg = sc["cfg"]
    _ssd_case(H, sc["L"][0], sc["R"][0], sc["kps2d"], cfg["window_size_depth_calculator"],
              cfg["search_x"], cfg["search_y"], 1)


# ----------------------------------------------------------------------- B2
def _klt_case(H, prev, cur, prev_pts, init, win):
    pl, cl = O.build_lk_pyramid(prev, win), O.build_lk_pyramid(cur, win)
    ref_pts, ref_st, ref_err = O.klt_track(pl, cl, prev_pts, init, win)
    gp, gc = H.build_lk_pyramid(dev(prev), win), H.build_lk_pyramid(dev(cur), win)
    cur_pts = dev(init.copy())
    _, st, err = H.klt_track(gp, gc, dev(prev_pts), cur_pts, win)
    st = st.cpu().numpy()
    assert np.array_equal(st, ref_st)
    # integer window sums => the float results are reproduced exactly
    assert np.array_equal(cur_pts.cpu().numpy(), ref_pts)
    assert np.array_equal(err.cpu().numpy(), ref_err)
    return ref_st


def test_klt_real_pair_bit_exact(H):
    left, right = util.real_pair()
    rng = np.random.RandomState(5)
    pts = np.stack([rng.uniform(-10, 760, 400), rng.uniform(-10, 490, 400)], 1).astype(np.float32)
    init = pts + rng.uniform(-3, 3, pts.shape).astype(np.float32) + np.float32([20, 0])
    st = _klt_case(H, left, right, pts, init.astype(np.float32), 31)
    assert st.sum() > 100
    _klt_case(H, left, right, pts, init.astype(np.float32), 35)


def test_klt_synthetic_motion(H):
    sc = util.scenario("euroc", 3, 1, 1)
    rng = np.random.RandomState(6)
    pts = sc["kps2d"]
    init = pts + rng.uniform(-2, 2, pts.shape).astype(np.float32)
    _klt_case(H, sc["L"][0], sc["L"][2], pts, init.astype(np.float32), 31)


def test_klt_small_image_fewer_levels(H):
    rng = np.random.RandomState(7)
    img = rng.randint(0, 256, (50, 70)).astype(np.uint8)
    img2 = np.roll(img, 1, axis=1)
    pts = np.stack([rng.uniform(0, 70, 40), rng.uniform(0, 50, 40)], 1).astype(np.float32)
    _klt_case(H, img, img2, pts, pts.copy(), 21)


# ------------------------------------------------------------------------ A
def _sia_inputs(sc, frame):
    cfg = sc["cfg"]
    nl = cfg["max_pyramid_levels"]
    return (O.build_pyramid(sc["L"][frame - 1], nl), O.build_pyramid(sc["L"][frame], nl),
            sc["kps2d"], sc["kps3d"], util.flags_of(sc["info"]))


@pytest.mark.parametrize("config,seed", [("tiny", 0), ("euroc", 0), ("euroc", 3), ("blender", 1)])
@pytest.mark.parametrize("exact", [False, True])
def test_sia_first_gradient_matches(H, config, seed, exact):
    """H = sum J^T J, b and the GN step of the first get_gradient on the coarsest level.
    exact=True is the product default: row-by-row accumulation like the reference
    (pose_estimator.cpp:399-403, :472-477); exact=False is the opt-in fast solver
    (svo_set_fast_solver), which sums J^T (sum g g^T) J per keypoint in a tree."""
    sc = util.scenario(config, 3, seed, 1)
    cfg = sc["cfg"]
    prev, cur, k2, k3, fl = _sia_inputs(sc, 1)
    level = cfg["max_pyramid_levels"] - 1
    guess = np.zeros(6, np.float32)
    Href, bref, sref = O.sia_gradient(prev[level], cur[level], level, k2, k3, fl, sc["cam"], guess)
    H.set_exact_pinv(exact)
    gp = [dev(x) for x in prev]
    gc = [dev(x) for x in cur]
    _, _, _, dbg = H.sparse_align(gp, gc, dev(k2), dev(k3), dev(fl), cam_of(cfg), dev(guess),
                                  dbg_level=level)
    H.set_exact_pinv(True)         # back to the default
    dbg = dbg.cpu().numpy()
    Hg, bg, sg = dbg[:36].reshape(6, 6), dbg[36:42], dbg[42:48]
    if exact:
        # same products added in the same order, same SVD: the oracle's bits
        assert np.array_equal(Hg, Href) and np.array_equal(bg, bref) and np.array_equal(sg, sref)
        return
    scale = np.sqrt(np.outer(np.diag(Href), np.diag(Href))) + 1e-20
    assert np.max(np.abs(Hg - Href) / scale) < 2e-5      # float sums in a different order
    assert np.max(np.abs(bg - bref)) < 2e-5 * np.max(np.abs(bref)) + 1e-3
    assert np.max(np.abs(sg - sref)) < 5e-3 * np.max(np.abs(sref)) + 1e-6


@pytest.mark.parametrize("config,seed,frame", [("tiny", 0, 1), ("tiny", 2, 1), ("euroc", 0, 1),
                                               ("euroc", 3, 1), ("blender", 1, 1), ("econ", 0, 1)])
def test_sia_pose_matches_fast_solver(H, config, seed, frame):
    """svo_handle_set_fast_solver: J^T (sum g g^T) J in a tree and LDL^T in double instead of the
    reference's row-by-row sums and float Jacobi-SVD inverse; same minimum, pose within 1e-4 m / rad."""
    sc = util.scenario(config, 3, seed, 1)
    cfg = sc["cfg"]
    prev, cur, k2, k3, fl = _sia_inputs(sc, frame)
    guess = np.zeros(6, np.float32)
    pref, cref, tref = O.sparse_align(prev, cur, k2, k3, fl, sc["cam"], guess)
    H.set_exact_pinv(False)
    pose, cost, trace, _ = H.sparse_align([dev(x) for x in prev], [dev(x) for x in cur], dev(k2),
                                          dev(k3), dev(fl), cam_of(cfg), dev(guess))
    assert np.max(np.abs(pose.cpu().numpy() - pref)) < 1e-4, (pose, pref)
    assert abs(float(cost.cpu()) - cref) < 2e-3 * max(cref, 1.0) + 2.0
    # the cost itself is summed in the reference's order in every mode: the first evaluation evaluation
    # of the coarsest level (same pose, same images) has the oracle's bits
    tr = hip_lib.trace_to_numpy(trace)
    top = cfg["max_pyramid_levels"] - 1
    assert tr[top]["initial_cost"] == tref[top]["initial_cost"]
    H.set_exact_pinv(True)         # back to the default


@pytest.mark.parametrize("config,seed,frame", [("tiny", 0, 1), ("tiny", 2, 1), ("euroc", 0, 1),
                                               ("euroc", 3, 1), ("blender", 1, 1), ("econ", 0, 1)])
def test_sia_pose_matches(H, config, seed, frame):
    """Default mode: row-by-row normal equations, Jacobi-SVD pseudo-inverse, sequential cost sums:
    the reference's iteration trace, level by level, and its pose bit for bit."""
    H.set_exact_pinv(True)
    sc = util.scenario(config, 3, seed, 1)
    cfg = sc["cfg"]
    prev, cur, k2, k3, fl = _sia_inputs(sc, frame)
    guess = np.zeros(6, np.float32)
    pref, cref, tref = O.sparse_align(prev, cur, k2, k3, fl, sc["cam"], guess)
    pose, cost, trace, _ = H.sparse_align([dev(x) for x in prev], [dev(x) for x in cur], dev(k2),
                                          dev(k3), dev(fl), cam_of(cfg), dev(guess))
    pose = pose.cpu().numpy()
    tr = hip_lib.trace_to_numpy(trace)
    # SURVEY §8d states 1e-4 m / 1e-4 rad; with every sum in reference order and the oracle's
    # sin / cos / hypot (include/svo_libm.h) the floats are the oracle's
    assert np.array_equal(pose, pref), (pose, pref)
    assert float(cost.cpu()) == cref
    for l in range(cfg["min_pyramid_level_pose_estimation"], cfg["max_pyramid_levels"]):
        assert tr[l]["n_gradient"] == tref[l]["n_gradient"], (l, tr[l], tref[l])
        assert tr[l]["n_cost"] == tref[l]["n_cost"], (l, tr[l], tref[l])
        assert tr[l]["n_accepted"] == tref[l]["n_accepted"]
        assert tr[l]["exit_small"] == tref[l]["exit_small"]
        assert tr[l]["initial_cost"] == tref[l]["initial_cost"] and tr[l]["final_cost"] == tref[l]["final_cost"]
    H.set_exact_pinv(True)         # back to the default


def test_sia_no_valid_patch_is_a_clean_exit(H):
    """All patches out of bounds: H = 0 -> pinv = 0 -> zero step -> pose unchanged."""
    sc = util.scenario("tiny", 3, 0, 1)
    cfg = sc["cfg"]
    prev, cur, k2, k3, fl = _sia_inputs(sc, 1)
    k2 = np.full_like(k2, -500.0)
    guess = np.array([0.01, 0, 0, 0, 0.002, 0], np.float32)
    pref, _, _ = O.sparse_align(prev, cur, k2, k3, fl, sc["cam"], guess)
    pose, _, _, _ = H.sparse_align([dev(x) for x in prev], [dev(x) for x in cur], dev(k2), dev(k3),
                                   dev(fl), cam_of(cfg), dev(guess))
    assert np.array_equal(pose.cpu().numpy(), pref)
    assert np.array_equal(pref, guess)


# ------------------------------------------------------------------ B1 + B3
@pytest.mark.parametrize("config,seed", [("tiny", 0), ("euroc", 0), ("euroc", 4), ("econ", 2), ("hd", 0)])
@pytest.mark.parametrize("exact", [False, True])
def test_reproj_gn_matches(H, config, seed, exact):
    """Cost and normal equations are summed in the reference's sequential order
    (pose_refinement.cpp:328-341, :393-395) in both modes, so the line search that stops on
    |dcost| < 1e-4 (:273) takes the oracle's path: the trace is asserted equal."""
    sc = util.scenario(config, 2 if config == "hd" else 3, seed, 1)
    cfg = sc["cfg"]
    rng = np.random.RandomState(8)
    k3, fl = sc["kps3d"], util.flags_of(sc["info"]).copy()
    true_pose = np.array([0.02, -0.01, 0.03, 0.004, -0.006, 0.002], np.float32)
    obs = O.project_keypoints(true_pose, k3, sc["cam"])
    tracked = (obs + rng.normal(0, 0.2, obs.shape)).astype(np.float32)
    tracked[::17] += 15.0                      # moved more than 9 px
    err = rng.uniform(0, 10, len(k3)).astype(np.float32)
    err[::13] = 30.0                           # occluded
    err[5] = np.inf
    start = np.zeros(6, np.float32)
    proj = O.project_keypoints(start, k3, sc["cam"])
    k2_ref, fl_ref = O.refine_merge(proj, fl, tracked, err)
    pref, cref, tref = O.reproj_gn(k2_ref, k3, fl_ref, sc["cam"], start)
    k2_g, fl_g = dev(proj.copy()), dev(fl.copy())
    H.set_exact_pinv(exact)
    pose, cost, trace = H.reproj_gn(k2_g, dev(k3), fl_g, cam_of(cfg), dev(start), dev(tracked), dev(err))
    H.set_exact_pinv(True)         # back to the default
    assert np.array_equal(fl_g.cpu().numpy(), fl_ref)            # flags: bit exact
    assert np.array_equal(k2_g.cpu().numpy(), k2_ref)
    if exact:
        assert np.array_equal(pose.cpu().numpy(), pref) and float(cost.cpu()) == cref
    assert np.max(np.abs(pose.cpu().numpy() - pref)) < 1e-4
    tr = hip_lib.trace_to_numpy(trace)[0]
    if exact:
        assert int(tr["n_gradient"]) == tref["n_gradient"], (tr, tref)
        assert int(tr["n_cost"]) == tref["n_cost"], (tr, tref)
        assert int(tr["n_accepted"]) == tref["n_accepted"]
    # (fast solver, exact=False: the LDL^T solve in double is not the reference's float SVD inverse, whose
    # error in the weak directions of J^T J is large; the steps differ, the minimum within 1e-4 does not)
    assert tr["initial_cost"] == tref["initial_cost"]            # same pose, sequential sum: same bits
    assert abs(float(cost.cpu()) - cref) < (1e-5 if exact else 1e-3)


# ------------------------------------------------------------------ C2 + D1
def test_depth_filter_update_matches(H):
    sc = util.scenario("euroc", 3, 0, 1)
    cfg = sc["cfg"]
    rng = np.random.RandomState(9)
    n = len(sc["kps3d"])
    k3 = sc["kps3d"]
    frame_pose = np.array([0.15, 0.12, 0.02, 0.01, -0.02, 0.005], np.float32)
    kf_pose = np.zeros((n, 6), np.float32)
    kf_pose[n // 2:] = np.array([0.01, 0.0, 0.0, 0.001, 0.0, 0.0], np.float32)
    k2 = O.project_keypoints(frame_pose, k3, sc["cam"]) + rng.normal(0, 0.3, (n, 2)).astype(np.float32)
    k2 = k2.astype(np.float32)
    ref2d = sc["kps2d"]
    disp = O.ssd_disparity(sc["L"][0], sc["R"][0], ref2d, 31, 60, 6, 1)
    disp[::11] += 6.0
    disp[3] = -1.0
    fl = util.flags_of(sc["info"]).copy()
    fl[::7] |= 1
    fl[::19] |= 2
    outl = rng.randint(0, 3, n).astype(np.int32)
    inl = rng.randint(0, 3, n).astype(np.int32)
    kx = sc["info"]["kf_inv_depth"].copy()
    kP = sc["info"]["kf_variance"].copy()
    o_ref, i_ref = O.outlier_check(k2, disp, sc["cam"], frame_pose, k3, kf_pose, outl, inl)
    k3_ref, o_ref2, kx_ref, kP_ref = O.update_kps3d(k2, k3, fl, sc["cam"], frame_pose, ref2d, kf_pose,
                                                    o_ref, kx, kP)
    g3, go, gi, gx, gP = dev(k3.copy()), dev(outl.copy()), dev(inl.copy()), dev(kx.copy()), dev(kP.copy())
    H.depth_filter_update(dev(k2), g3, dev(fl), cam_of(cfg), dev(frame_pose), dev(disp), dev(k3),
                          dev(ref2d), dev(kf_pose), go, gi, gx, gP)
    assert np.array_equal(go.cpu().numpy(), o_ref2)              # counters: bit exact
    assert np.array_equal(gi.cpu().numpy(), i_ref)
    # same expressions, IEEE float, no contraction: expected equal; 1e-5 rel is the stated bound
    assert np.allclose(gx.cpu().numpy(), kx_ref, rtol=1e-5, atol=0)
    assert np.allclose(gP.cpu().numpy(), kP_ref, rtol=1e-5, atol=0)
    assert np.allclose(g3.cpu().numpy(), k3_ref, rtol=1e-5, atol=1e-6)
    assert (np.abs(k3_ref - k3).max(axis=1) > 0).sum() > n // 4   # the update really ran


# ------------------------------------------------ edge cases and size-independent properties
def test_empty_and_single_keypoint_inputs(H):
    """n = 0 and n = 1 through every stage entry (the reference handles an empty set by
    printing 'This should never happen', pose_estimator.cpp:247-258, and carries on)."""
    sc = util.scenario("tiny", 3, 0, 1)
    cfg = sc["cfg"]
    cam = cam_of(cfg)
    nl = cfg["max_pyramid_levels"]
    prev, cur = O.build_pyramid(sc["L"][0], nl), O.build_pyramid(sc["L"][1], nl)
    gp, gc = [dev(x) for x in prev], [dev(x) for x in cur]
    guess = np.array([0.01, -0.02, 0.0, 0.001, 0.0, 0.002], np.float32)
    for n in (0, 1):
        k2 = sc["kps2d"][:n].copy().reshape(n, 2)
        k3 = sc["kps3d"][:n].copy().reshape(n, 3)
        fl = np.zeros(n, np.uint32)
        pref, _, _ = O.sparse_align(prev, cur, k2, k3, fl, sc["cam"], guess)
        d2 = torch.zeros((max(n, 1), 2), dtype=torch.float32, device="cuda")[:n]
        d3 = torch.zeros((max(n, 1), 3), dtype=torch.float32, device="cuda")[:n]
        dfl = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda")[:n]
        if n:
            d2.copy_(torch.from_numpy(k2)); d3.copy_(torch.from_numpy(k3))
        pose, _, _, _ = H.sparse_align(gp, gc, d2, d3, dfl, cam, dev(guess))
        if n == 0:
            assert np.array_equal(pose.cpu().numpy(), pref)      # H = 0 -> zero step -> pose unchanged
        else:
            # one patch gives a rank-2 J^T J: the reference's float SVD inverse amplifies rounding
            # noise by ~1e7 there, so only sanity is checked
            assert np.all(np.isfinite(pose.cpu().numpy()))
        disp = H.ssd_disparity(dev(sc["L"][1]), dev(sc["R"][1]), d2, 21, 30, 4, 1)
        assert np.array_equal(disp.cpu().numpy(), O.ssd_disparity(sc["L"][1], sc["R"][1], k2, 21, 30, 4, 1))
        lk = H.build_lk_pyramid(dev(sc["L"][0]), 21)
        cur_pts = d2.clone()
        _, st, err = H.klt_track(lk, lk, d2, cur_pts, 21)
        assert st.shape[0] == n
        p2, _, _ = H.reproj_gn(d2.clone(), d3, dfl.clone(), cam, dev(guess))
        pr, _, _ = O.reproj_gn(k2, k3, fl, sc["cam"], guess)
        if n == 0:
            assert np.array_equal(p2.cpu().numpy(), pr)
        else:
            assert np.all(np.isfinite(p2.cpu().numpy()))


def test_identity_properties_full_size(H):
    """Size-independent properties at the full 752x480 / 1920x1080 sizes: an image
    tracked against itself stays put with zero error, identical stereo images give
    the clamped disparity 0.5, aligning a frame with itself keeps the pose."""
    for config in ("euroc", "hd"):
        cfg = dict(synth.CONFIGS[config])
        rng = np.random.RandomState(12)
        from scipy import ndimage
        img = ndimage.gaussian_filter(rng.uniform(0, 255, (cfg["height"], cfg["width"])), 2.0)
        img = ((img - img.min()) / (img.max() - img.min()) * 255).astype(np.uint8)
        n = 300
        pts = np.stack([rng.uniform(40, cfg["width"] - 100, n), rng.uniform(40, cfg["height"] - 40, n)], 1).astype(np.float32)
        g = dev(img)
        win = cfg["window_size_opt_flow"]
        lk = H.build_lk_pyramid(g, win)
        cur_pts = dev(pts.copy())
        _, st, err = H.klt_track(lk, lk, dev(pts), cur_pts, win)
        ok = st.cpu().numpy() > 0
        assert ok.mean() > 0.9
        assert np.max(np.abs(cur_pts.cpu().numpy()[ok] - pts[ok])) < 1e-3
        assert np.all(err.cpu().numpy()[ok] == 0)
        disp = H.ssd_disparity(g, g, dev(pts), cfg["window_size_depth_calculator"], cfg["search_x"],
                               cfg["search_y"], 1).cpu().numpy()
        assert np.all(disp == 0.5)
        nl = cfg["max_pyramid_levels"]
        pyr = H.build_pyramid(g, nl)
        k3 = np.stack([(pts[:, 0] - cfg["cx"]) / cfg["fx"] * 3, (pts[:, 1] - cfg["cy"]) / cfg["fy"] * 3,
                       np.full(n, 3.0)], 1).astype(np.float32)
        zero = np.zeros(6, np.float32)
        pose, cost, _, _ = H.sparse_align(pyr, pyr, dev(pts), dev(k3), dev(np.zeros(n, np.uint32)),
                                          cam_of(cfg), dev(zero))
        assert np.max(np.abs(pose.cpu().numpy())) < 1e-5 and float(cost.cpu()) < 1.0
