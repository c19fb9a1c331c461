#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/dp2_loop
timeout 1100 python -m pytest tests -x -q -m gpu --durations=45 > gpurun_out/gpu_tests_r06a.log 2>&1
echo "pytest rc $?" >> gpurun_out/gpu_tests_r06a.log
DP2_WATCHDOG_S=60 timeout 1500 python tools/dp2_loop.py 25 4 2 1 > gpurun_out/dp2_loop_r06a.log 2>&1
tail -5 gpurun_out/gpu_tests_r06a.log; tail -3 gpurun_out/dp2_loop_r06a.log
