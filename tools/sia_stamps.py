"""Diagnostic (not a test, not the product): builds libsvo_hip with -DSVO_SIA_STAMPS and prints
where one sparse-alignment launch spends its cycles (s_memtime on thread 0)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stereo-svo-slam_amd", "csrc")
THREADS = os.environ.get("SIA_THREADS", "1024")
out = os.path.join("/tmp", f"libsvo_hip_stamps_{THREADS}.so")
srcs = [os.path.join(CSRC, f) for f in ("svo_capi.hip svo_ctx.hip pyramid.hip sia.hip klt.hip reproj.hip depth.hip keyframe.hip").split()]
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC",
                       "-std=c++17", "-DSVO_SIA_STAMPS", f"-DSVO_SIA_THREADS={THREADS}", "-shared", "-o", out] + srcs)
os.environ["SVO_HIP_LIB"] = out
sys.path[:0] = [ROOT, os.path.join(ROOT, "stereo-svo-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import oracle_py as O
from stereo_svo_slam_amd import hip_lib
import util
sc = util.scenario("euroc", 3, 0, 1)
cfg = sc["cfg"]; nl = cfg["max_pyramid_levels"]
prev, cur = O.build_pyramid(sc["L"][0], nl), O.build_pyramid(sc["L"][1], nl)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
H = hip_lib.Handle(0, 448)
args = ([d(x) for x in prev], [d(x) for x in cur], d(sc["kps2d"]), d(sc["kps3d"]), d(util.flags_of(sc["info"])),
        hip_lib.CameraSettings.from_dict(cfg), d(np.zeros(6, np.float32)))
for it in range(3):
    pose, cost, trace, dbg = H.sparse_align(*args, dbg_level=0)
    torch.cuda.synchronize()
s = dbg.cpu().numpy()[:12]
tr = hip_lib.trace_to_numpy(trace)
ng = sum(int(t["n_gradient"]) for t in tr); nc = sum(int(t["n_cost"]) for t in tr)
names = ["cost:sync0", "cost:pose_mats", "cost:sync+project", "cost:sync", "cost:taps", "cost:reduce", "n_cost",
         "grad:to_reduce_end", "grad:solve", "n_grad", "kernel_total", "-"]
print("n_grad", ng, "n_cost", nc)
for n_, v in zip(names, s):
    per = v / max(s[6], 1) if n_.startswith("cost") else (v / max(s[9], 1) if n_.startswith("grad") else v)
    print(f"{n_:22s} total {v:12.0f}  per call {per:10.0f} cycles")
