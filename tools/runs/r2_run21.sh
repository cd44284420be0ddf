#!/bin/bash
B="--no-cpu-baseline --no-extras --repeats 1"
for v in "SVO_SIA_MODE=1 SVO_SIA_WAVES=2" "SVO_SIA_MODE=2 SVO_SIA_WAVES=1" "SVO_SIA_MODE=2 SVO_SIA_WAVES=2" "SVO_SIA_MODE=1 SVO_SIA_WAVES=1" "SVO_SIA_MODE=0 SVO_SIA_WAVES=1" "SVO_SIA_MODE=0 SVO_SIA_WAVES=2"; do
  echo "== $v"; env $v timeout -k 10 300 python bench.py $B 2>/dev/null | cut -c1-120
done
A="--steps 10 --repeats 1 --no-cpu-baseline --no-extras"
bash tools/profile_timeline.sh r2_tl8 $A | tail -24
