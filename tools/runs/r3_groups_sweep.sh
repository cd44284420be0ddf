#!/bin/bash
# sequence count / group count sweep on the round-3 workload
A="--no-extras --no-cpu-baseline --repeats 1 --steps 40"
run() { # queues groups seqs
  GPU_MAX_HW_QUEUES=$1 SVO_GROUPS=$2 timeout -k 10 300 python bench.py $A --seqs $3 > /tmp/g.json 2> /tmp/g.err || { tail -2 /tmp/g.err; return; }
  python - "$1" "$2" "$3" <<'PY'
import json, sys
j = json.load(open("/tmp/g.json"))
print("queues %s groups %s seqs %s: %.0f fps, %.2f ms/step" % (*sys.argv[1:4], j["value"], j["ms_per_step"]),
      {k[:6]: round(v, 2) for k, v in j["roofline"]["stage_ms_per_launch"].items()}, flush=True)
PY
}
run 12 8 2048
run 12 11 2816
run 20 16 2048
run 20 16 4096
run 12 8 4096
run 12 4 2048
