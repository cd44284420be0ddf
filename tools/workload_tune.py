"""Diagnostic: keyframe rate / keypoints per frame / GN calls of the HIP tracker on the closed-loop
synthetic workload, for a few motion scales. Usage: workload_tune.py [config] [n_seq] [loop] [scales...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "stereo-svo-slam_amd")]
import numpy as np, torch
from stereo_svo_slam_amd import synth
from stereo_svo_slam_amd.stereo_slam import StereoSlamBatch

config = sys.argv[1] if len(sys.argv) > 1 else "euroc"
n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 16
loop = int(sys.argv[3]) if len(sys.argv) > 3 else 192
scales = [float(v) for v in sys.argv[4:]] or [0.5, 1.0, 1.5]
dev = torch.device("cuda", 0)
cfg = dict(synth.CONFIGS[config])
for sc in scales:
    t0 = time.perf_counter()
    L, R = [], []
    for s in range(n_seq):
        scene = synth.Scene(s % 8, dev)
        poses = synth.loop_trajectory(loop, s, sc)
        seeds = 7919 * (s + 1) + 2 * np.arange(loop)
        L.append(synth.render_frames_gpu(scene, cfg, poses, False, 1.0, seeds))
        R.append(synth.render_frames_gpu(scene, cfg, poses, True, 1.0, seeds + 1))
    torch.cuda.synchronize()
    t_render = time.perf_counter() - t0
    os.environ["SVO_GROUPS"] = "1"
    slam = StereoSlamBatch(cfg, cfg["width"], cfg["height"], n_seq)
    n_steps = loop + loop // 2
    kf_at = [[] for _ in range(n_seq)]
    nk = []
    for k in range(n_steps):
        off = [(s * 11) % loop for s in range(n_seq)]
        slam.new_images_packed(slam.pack_images([L[s][(off[s] + k) % loop] for s in range(n_seq)],
                                                [R[s][(off[s] + k) % loop] for s in range(n_seq)], [k / 20.0] * n_seq, borrow=True))
        for s in range(n_seq):
            st = slam.stats(s)
            if st.is_keyframe:
                kf_at[s].append(k)
            nk.append(st.n_keypoints)
    t = slam.totals()
    gaps = np.concatenate([np.diff(x) for x in kf_at if len(x) > 1]) if any(len(x) > 1 for x in kf_at) else np.array([0])
    d = np.abs(np.diff(synth.loop_trajectory(loop, 0, sc), axis=0))
    print(f"scale {sc}: keyframe rate {t.keyframes / t.frames:.4f} (gap median {np.median(gaps):.0f}, min {gaps.min()}, max {gaps.max()}), "
          f"kps/frame {t.keypoints / t.frames:.1f} (min {min(nk)}, max {max(nk)}), grad/frame {t.gn_gradient_calls / t.frames:.2f}, "
          f"cost/frame {t.gn_cost_calls / t.frames:.2f}; per-frame motion max {d[:, :3].max() * 100:.1f} cm, {np.degrees(d[:, 3:].max()):.2f} deg; "
          f"render {t_render:.1f} s for {2 * n_seq * loop} images", flush=True)
    slam.close()
