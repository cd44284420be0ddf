#!/bin/bash
# default bench: baseline library (commit 6ec93d4 kernels + per-set hipMalloc) vs the current one vs 32-keypoint staging
A="--no-extras --repeats 1 --steps 60"
export SVO_HOST_TIMING=1
for rep in 1 2; do
  for lib in build_ab/libsvo_hip_base.so stereo-svo-slam_amd/csrc/libsvo_hip.so build_ab/libsvo_hip_stg32.so; do
    SVO_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 400 python3 bench.py --no-cpu-baseline $A > /tmp/ab.json 2> /tmp/ab.err || { tail -3 /tmp/ab.err; exit 1; }
    python3 -c "
import json,sys; j=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('%-28s %8.0f fps %.3f ms/step '%(sys.argv[1], j['value'], j['ms_per_step']), {k[:6]:round(v,3) for k,v in j['roofline']['stage_ms_per_launch'].items()}, flush=True)" $lib
    grep "svo host" /tmp/ab.err | tail -2
  done
done
