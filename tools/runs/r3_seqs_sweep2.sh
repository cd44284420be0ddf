#!/bin/bash
# end-of-round check: frames/s against sequences in flight (groups of 256) with the final kernels
for S in 2048 2560 2816 2048; do
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extras --repeats 2 --steps 60 --seqs $S > /tmp/ss.json 2> /tmp/ss.err || { tail -3 /tmp/ss.err; exit 1; }
  python3 -c "
import json,sys; j=json.loads(open('/tmp/ss.json').read().strip().splitlines()[-1]); print('seqs', sys.argv[1], 'groups', j['config']['sequence_groups'], [round(v/1e3,1) for v in j['config']['repeats_fps']], 'K frames/s', round(j['ms_per_step'],2), 'ms/step', flush=True)" $S
done
