#!/bin/bash
# KLT template ring of 4 against 8 keyframes per sequence (same library, interleaved)
for rep in 1 2 3; do
  for KF in 4 8; do
    SVO_KLT_CACHE_KF=$KF timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extras --repeats 2 --steps 60 > /tmp/ring.json 2> /tmp/ring.err || { tail -3 /tmp/ring.err; exit 1; }
    python3 -c "
import json,sys; j=json.loads(open('/tmp/ring.json').read().strip().splitlines()[-1]); print('ring', sys.argv[1], [round(v/1e3,1) for v in j['config']['repeats_fps']], 'K frames/s; klt stage', round(j['roofline']['stage_ms_per_launch']['klt'],3), 'image sets', j['config'].get('image_sets_allocated'), 'keyframes', j['config'].get('keyframes_created'), flush=True)" $KF
  done
done
