#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== tests"; timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "grouped_wgrad" 2>&1 | tail -15
for v in 0 1; do
  bash tools/kprof.sh wd$v "gemm_wgrad" VPF_WGROUP_DMA=$v -- wstack > /dev/null 2>&1
done
echo "== A/B step"; bash tools/ab.sh "VPF_WGROUP_DMA=0" "VPF_WGROUP_DMA=1" 3
} > gpurun_out/r05_dma1.txt 2>&1
cat gpurun_out/r05_dma1.txt | grep -v amdgpu.ids
