#!/bin/bash
# what the driver runs at round end, on the final commit: GPU tests, smoke, default bench (no CPU leg here)
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r02_final_tests.log 2>&1; tail -2 gpurun_out/r02_final_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-200
