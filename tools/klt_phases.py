"""Diagnostic (not a test, not the product): runs bench.py's default command on a libsvo_hip built with
-DSVO_KLT_PHASES (build_ab/libsvo_hip_kltphases.so) and prints where the wavefronts of klt_track_kernel
spend their cycles (s_memtime of thread 0, summed over all wavefronts of the run).
Usage: klt_phases.py [bench.py arguments]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SVO_HIP_LIB", os.path.join(ROOT, "build_ab", "libsvo_hip_kltphases.so"))
sys.path[:0] = [ROOT, os.path.join(ROOT, "stereo-svo-slam_amd")]
import bench
from stereo_svo_slam_amd import hip_lib
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extras", "--repeats", "1", "--steps", "60"] + sys.argv[1:]
bench.main()
out = (C.c_ulonglong * 16)()
assert hip_lib.lib().svo_debug_klt_phases(out) == 0
v = list(out)
waves, total = max(v[8], 1), max(sum(v[:5]) + v[5] + v[6] + v[7] + v[11] + v[12] + v[13], 1)
names = ["prologue", "template (requested or built)", "search tile staged (+ loads in flight)", "iterations", "error pass"]
print(f"klt_track_kernel: {waves} wavefronts, {v[10] / waves:.1f} iterations and {v[9] / waves:.2f} tile stagings each,"
      f" {total / waves:.0f} ticks of s_memtime per wavefront", file=sys.stderr)
for n_, c in zip(names, v[:5]):
    print(f"  {n_:40s} {c / waves:9.0f} ticks  {100.0 * c / total:5.1f} %", file=sys.stderr)
if any(v[i] for i in (5, 6, 7, 11, 12, 13)):        # builds with the diagnostic waits: the two memory phases split up
    for n_, i in (("level: image views and addresses", 5), ("wait: flag, header, template, tile", 6),
                  ("(builds before the loads were grouped) header", 7), ("(before ...) template", 11),
                  ("iteration 0 up to a staging", 12), ("tile loads + LDS stores (restagings only)", 13)):
        print(f"    {n_:38s} {v[i] / waves:9.0f} ticks", file=sys.stderr)
