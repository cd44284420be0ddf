"""Diagnostic: build libsvo_hip variants with extra -D flags and print the per-stage times of
bench.py for each (SVO_HIP_LIB override). Usage: variant_bench.py "name:-DX=1 -DY=2" ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stereo-svo-slam_amd", "csrc")
srcs = [os.path.join(CSRC, f) for f in "svo_capi.hip svo_ctx.hip pyramid.hip sia.hip klt.hip reproj.hip depth.hip keyframe.hip".split()]
extra = os.environ.get("BENCH_ARGS", "--seqs 64 --steps 30 --warmup 3 --no-cpu-baseline").split()
for spec in sys.argv[1:]:
    name, flags = spec.split(":", 1)
    out = f"/tmp/libsvo_hip_{name}.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC",
                           "-std=c++17", "-shared", "-o", out] + flags.split() + srcs)
    env = dict(os.environ, SVO_HIP_LIB=out)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        st = d["roofline"]["stage_ms_per_launch"]
        print(f"{name:14s} fps {d['value']:9.0f} step {d['ms_per_step']:.3f} ms | " + " ".join(f"{k[:6]}={v:.3f}" for k, v in st.items()), flush=True)
    except Exception as e:
        print(name, "FAILED", e, r.stderr[-500:])
