#!/bin/bash
# C3: compaction / merge kernels with 256 instead of 1024 threads (parity legs on)
for T in "0 0" "256 0" "256 256"; do
  set -- $T
  [ "$1" != 0 ] && export SVO_COMPACT_THREADS=$1
  [ "$2" != 0 ] && export SVO_MERGE_THREADS=$2
  timeout -k 10 500 python3 bench.py --config hd --steps 12 --warmup 4 --repeats 1 --no-extras > /tmp/hdt.json 2> /tmp/hdt.err || { tail -3 /tmp/hdt.err; exit 1; }
  python3 -c "
import json,sys; j=json.loads(open('/tmp/hdt.json').read().strip().splitlines()[-1]); print('compact/merge threads', sys.argv[1], sys.argv[2], '%8.0f fps %.3f ms/step parity %s over %s'%(j['value'], j['ms_per_step'], j['parity_max_abs_pose_diff'], j['parity_sequences_compared']), {k[:6]:round(v,3) for k,v in j['roofline']['stage_ms_per_launch'].items()}, flush=True)" $1 $2
done
