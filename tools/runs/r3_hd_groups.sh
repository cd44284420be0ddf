#!/bin/bash
# C3 (1920x1080): frames/s against the number of sequence groups the 256 sequences are cut into
for G in 1 2 4 8; do
  SVO_GROUPS=$G timeout -k 10 400 python3 bench.py --config hd --seqs 256 --steps 12 --warmup 4 --repeats 1 --no-extras --no-cpu-baseline > /tmp/hdg.json 2> /tmp/hdg.err || { tail -3 /tmp/hdg.err; exit 1; }
  python3 -c "
import json,sys; j=json.loads(open('/tmp/hdg.json').read().strip().splitlines()[-1]); print('groups', sys.argv[1], '%8.0f fps %.3f ms/step '%(j['value'], j['ms_per_step']), {k[:6]:round(v,3) for k,v in j['roofline']['stage_ms_per_launch'].items()}, flush=True)" $G
done
