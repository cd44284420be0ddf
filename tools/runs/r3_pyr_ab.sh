#!/bin/bash
# pyramid kernels A/B: alone (one sequence group of 256) and in the default bench; then the hd config
A="--no-extras --no-cpu-baseline --repeats 1"
pick() { python - "$1" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[1], "fps %.0f" % j["value"], "pyr ms/launch %.4f" % j["roofline"]["stage_ms_per_launch"]["images+pyramids"],
      "pyr hbm frac %.3f" % j["roofline"]["pyramids_hbm"]["frac"], {k: round(v, 3) for k, v in j["roofline"]["stage_ms_per_launch"].items()})
PY
}
for k in stream tile; do
  SVO_PYR_KERNEL=$k SVO_GROUPS=1 timeout -k 10 200 python bench.py $A --seqs 256 --loops 16 --steps 30 > gpurun_out/r3_pyr_alone_$k.json 2> gpurun_out/r3_pyr_alone_$k.err; pick gpurun_out/r3_pyr_alone_$k.json
  SVO_PYR_KERNEL=$k timeout -k 10 200 python bench.py $A > gpurun_out/r3_pyr_full_$k.json 2> gpurun_out/r3_pyr_full_$k.err; pick gpurun_out/r3_pyr_full_$k.json
done
timeout -k 10 400 python bench.py --config hd --seqs 128 --steps 12 --warmup 4 $A > gpurun_out/r3_hd_a.json 2> gpurun_out/r3_hd_a.err; pick gpurun_out/r3_hd_a.json
python - <<'PY'
import json
j = json.load(open("gpurun_out/r3_hd_a.json"))
print("hd", j["value"], j["patches_per_frame"], j["keyframe_rate"], j["config"]["gn_gradient_calls_per_frame"], j["config"]["gn_cost_calls_per_frame"], j["setup_s"], j["wall_s"])
PY
