#!/bin/bash
T="tests/test_tracker_gpu.py::test_groups_and_pipelined_submit_equal_lockstep"
for i in 1 2 3 4 5 6; do timeout -k 10 200 python -m pytest $T -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|^E " | head -4; done
