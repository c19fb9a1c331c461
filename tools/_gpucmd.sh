#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3 4 5; do
  s=$(date +%s)
  python -m pytest tests/ -x -q -m gpu > gpurun_out/soak_$i.log 2>&1
  echo "suite run $i: rc $? $(( $(date +%s) - s )) s: $(grep -E "passed|failed" gpurun_out/soak_$i.log | tail -1)" | tee -a gpurun_out/r06_gpu_tests_soak.log
done
