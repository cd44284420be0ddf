#!/bin/bash
# more hardware queues: frames/s against sequences in flight (groups of 256)
export GPU_MAX_HW_QUEUES=20
for S in 2816 3584 4096 2816; do
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extras --repeats 2 --steps 50 --seqs $S > /tmp/ss.json 2> /tmp/ss.err || { tail -3 /tmp/ss.err; exit 1; }
  python3 -c "
import json,sys; j=json.loads(open('/tmp/ss.json').read().strip().splitlines()[-1]); print('20 queues, seqs', sys.argv[1], 'groups', j['config']['sequence_groups'], [round(v/1e3,1) for v in j['config']['repeats_fps']], 'K frames/s', round(j['ms_per_step'],2), 'ms/step', flush=True)" $S
done
