#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/vpf_prof/tl
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/vpf_prof/tl -o tl -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernels --no-variants > /tmp/vpf_prof/tl/log.txt 2>&1
f=$(find /tmp/vpf_prof/tl -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f 2 > gpurun_out/r05_kernel_timeline_c2.txt 2>&1
tail -12 gpurun_out/r05_kernel_timeline_c2.txt
