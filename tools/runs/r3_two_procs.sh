#!/bin/bash
# is the GPU saturated, or is one process's structure (streams, host threads) the limit?
# one process with 2048 sequences, then two such processes at the same time
A="--no-extras --no-cpu-baseline --repeats 1 --steps 60"
python bench.py $A > /tmp/one.json 2>/dev/null
python bench.py $A > /tmp/a.json 2>/dev/null &
PA=$!
python bench.py $A > /tmp/b.json 2>/dev/null &
PB=$!
wait $PA; wait $PB
python - <<'PY'
import json
one, a, b = (json.load(open(f"/tmp/{n}.json")) for n in ("one", "a", "b"))
print("one process: %.0f fps" % one["value"])
print("two processes at once: %.0f + %.0f = %.0f fps" % (a["value"], b["value"], a["value"] + b["value"]))
PY
