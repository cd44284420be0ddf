#!/bin/bash
# frames/s of consecutive repeats of the timed region (does sustained running slow down?), by size of the KLT template ring
for KF in 4 8; do
  SVO_KLT_CACHE_KF=$KF timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extras --repeats 6 --steps 80 > /tmp/rep.json 2> /tmp/rep.err || { tail -3 /tmp/rep.err; exit 1; }
  python3 -c "
import json,sys; j=json.loads(open('/tmp/rep.json').read().strip().splitlines()[-1]); print('template ring of', sys.argv[1], 'keyframes:', [round(v/1e3,1) for v in j['config']['repeats_fps']], 'K frames/s; klt stage', round(j['roofline']['stage_ms_per_launch']['klt'],3), flush=True)" $KF
done
