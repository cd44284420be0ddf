"""Diagnostic (not a test, not the product): loads a libsvo_hip built with -DSVO_SIA_STAMPS
(tools/build_variants.sh stamps -> build_ab/libsvo_hip_stamps.so) and prints where one
sparse-alignment launch spends its cycles (s_memtime on thread 0), for a lone sequence
(workgroup shape by keypoint count) and, with `batch`, for the one-wave shape of batched launches.
Usage: sia_stamps.py [config] [exact]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SVO_HIP_LIB", os.path.join(ROOT, "build_ab", "libsvo_hip_stamps.so"))
sys.path[:0] = [ROOT, os.path.join(ROOT, "stereo-svo-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import oracle_py as O
from stereo_svo_slam_amd import hip_lib
import util
config = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] not in ("fast",) else "euroc"
exact = "fast" not in sys.argv[1:]
sc = util.scenario(config, 3, 0, 1)
cfg = sc["cfg"]; nl = cfg["max_pyramid_levels"]
prev, cur = O.build_pyramid(sc["L"][0], nl), O.build_pyramid(sc["L"][1], nl)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
H = hip_lib.Handle(0, 448)
H.set_exact_pinv(exact)
args = ([d(x) for x in prev], [d(x) for x in cur], d(sc["kps2d"]), d(sc["kps3d"]), d(util.flags_of(sc["info"])),
        hip_lib.CameraSettings.from_dict(cfg), d(np.zeros(6, np.float32)))
for it in range(3):
    pose, cost, trace, dbg = H.sparse_align(*args, dbg_level=0)
    torch.cuda.synchronize()
s = dbg.cpu().numpy()[:12]
tr = hip_lib.trace_to_numpy(trace)
ng = sum(int(t["n_gradient"]) for t in tr); nc = sum(int(t["n_cost"]) for t in tr)
names = ["levels: image + records", "cost: pose_mats", "cost: keypoints", "cost: ordered sum", "n_cost",
         "grad: pose+keypoints", "grad: reduce", "grad: solve", "n_grad", "kernel_total"]
print(f"{config}: n = {len(sc['kps2d'])}, exact = {exact}, n_grad {ng}, n_cost {nc}")
for n_, v in zip(names, s):
    per = v / max(s[4], 1) if n_.startswith("cost") else (v / max(s[8], 1) if n_.startswith("grad") else v)
    print(f"{n_:26s} total {v:12.0f}  per call {per:10.0f} cycles (100 MHz ticks x clock ratio)")
