#!/bin/bash
A="--no-extras --no-cpu-baseline --repeats 1 --steps 60"
run() { # label env...
  L=$1; shift
  env "$@" timeout -k 10 300 python bench.py $A > /tmp/m.json 2> /tmp/m.err || { tail -2 /tmp/m.err; return; }
  python - "$L" <<'PY'
import json, sys
j = json.load(open("/tmp/m.json"))
print("%-26s %.0f fps, %.2f ms/step" % (sys.argv[1], j["value"], j["ms_per_step"]), {k[:6]: round(v, 2) for k, v in j["roofline"]["stage_ms_per_launch"].items()}, flush=True)
PY
}
run "default (mode 2, 1 wave)" X=1
run "mode 1" SVO_SIA_MODE=1
run "mode 0" SVO_SIA_MODE=0
run "mode 2, 2 waves" SVO_SIA_WAVES=2
run "default again" X=1
