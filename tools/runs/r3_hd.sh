#!/bin/bash
# BASELINE config 3 (1920x1080, 43x24 grid, 5-level SIA pyramid): 512 sequences in eight groups of 64, CPU legs on, kernel stats
S=$(date +%s)
timeout -k 10 900 python bench.py --config hd --seqs 512 --steps 16 --warmup 4 --repeats 2 --no-extras > gpurun_out/r03_c3_hd_bench.json 2> gpurun_out/r03_c3_hd_bench.err; echo "rc=$? $(( $(date +%s) - S )) s"; tail -2 gpurun_out/r03_c3_hd_bench.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r03_c3_hd_bench.json"))
print("fps", j["value"], "patches", j["patches_per_frame"], "kf rate", j["keyframe_rate"], "gn ms/iter", j["gn_ms_per_iter"],
      "cpu", j["cpu_one_core_fps"], j["cpu_all_cores_fps"], "parity", j["parity_max_abs_pose_diff"], j["parity_sequences_compared"])
print({k: round(v, 3) for k, v in j["roofline"]["stage_ms_per_launch"].items()}, j["roofline"]["pyramids_hbm"]["frac"])
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/rp_hd
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_hd -- python3 bench.py --config hd --seqs 512 --steps 8 --warmup 4 --repeats 1 --no-extras --no-cpu-baseline > gpurun_out/r03_c3_hd_bench_under_rocprof.json 2> /tmp/hd_kt.err
find /tmp/rp_hd -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03_c3_hd_kernel_stats.csv \;
head -12 gpurun_out/r03_c3_hd_kernel_stats.csv | cut -c1-140
# the pyramid stage alone: one sequence group
SVO_GROUPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_hd1 -- python3 bench.py --config hd --seqs 256 --steps 6 --warmup 2 --repeats 1 --no-extras --no-cpu-baseline > /dev/null 2> /tmp/hd_kt1.err
find /tmp/rp_hd1 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03_c3_hd_kernel_stats_1group.csv \;
grep -i "pyr_stream" gpurun_out/r03_c3_hd_kernel_stats_1group.csv | cut -c1-160
