#!/bin/bash
# C3 (1920x1080): frames/s against the number of sequences in flight (groups of 64 - 128)
for SG in "256 4" "512 8" "1024 8" "1024 11"; do
  set -- $SG
  SVO_GROUPS=$2 timeout -k 10 500 python3 bench.py --config hd --seqs $1 --steps 12 --warmup 4 --repeats 1 --no-extras --no-cpu-baseline > /tmp/hds.json 2> /tmp/hds.err || { tail -3 /tmp/hds.err; exit 1; }
  python3 -c "
import json,sys; j=json.loads(open('/tmp/hds.json').read().strip().splitlines()[-1]); print('seqs', sys.argv[1], 'groups', sys.argv[2], '%8.0f fps %.3f ms/step setup %.0f s '%(j['value'], j['ms_per_step'], j['setup_s']), {k[:6]:round(v,3) for k,v in j['roofline']['stage_ms_per_launch'].items()}, flush=True)" $1 $2
done
