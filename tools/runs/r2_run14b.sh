#!/bin/bash
T="tests/test_tracker_gpu.py::test_groups_and_pipelined_submit_equal_lockstep"
for i in 1 2; do timeout -k 10 200 python -m pytest $T -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2; done
for i in 1 2; do SVO_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/libsvo_hip_acc1.so timeout -k 10 200 python -m pytest $T -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2; done
timeout -k 10 600 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect $T > gpurun_out/r2_t14b.log 2>&1; tail -3 gpurun_out/r2_t14b.log
bash tools/runs/r2_run15.sh
