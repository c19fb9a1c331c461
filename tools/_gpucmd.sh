#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.txt
timeout 3000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_tests.log 2>&1
tail -5 gpurun_out/r05_gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
