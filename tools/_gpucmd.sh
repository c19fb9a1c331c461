#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export VPF_SCRATCH=/tmp/vpf_prof; mkdir -p $VPF_SCRATCH
rm -f gpurun_out/parity_report.txt
s=$(date +%s)
timeout 1100 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/r06_gpu_tests.log 2>&1
echo "pytest rc $? wall $(( $(date +%s) - s )) s" >> gpurun_out/r06_gpu_tests.log
tail -3 gpurun_out/r06_gpu_tests.log
cp gpurun_out/parity_report.txt gpurun_out/r06_parity_report.txt
bash tools/collect_step_bytes.sh r06 c2 2>&1 | tail -4
mkdir -p profiles_tmp; cp gpurun_out/r06_step_bytes.json profiles/r06_step_bytes.json
bash tools/collect_step_issue.sh r06 c2 2>&1 | tail -3
bash tools/collect_profiles.sh r06 2>&1 | tail -4
python3 bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
tail -c 600 gpurun_out/r06_bench_default.json
