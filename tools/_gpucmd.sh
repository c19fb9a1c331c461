#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r06_gpu_tests_3x.log
for i in 1 2 3 4; do
  s=$(date +%s)
  rm -f gpurun_out/parity_report.txt
  python -m pytest tests/ -x -q -m gpu > gpurun_out/suite_$i.log 2>&1
  echo "suite run $i: rc $? $(( $(date +%s) - s )) s: $(grep -E "passed|failed" gpurun_out/suite_$i.log | tail -1)" | tee -a gpurun_out/r06_gpu_tests_3x.log
done
cp gpurun_out/parity_report.txt gpurun_out/r06_parity_report.txt
