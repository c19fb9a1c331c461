#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== tests"; timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "grouped_wgrad" 2>&1 | tail -5
bash tools/kprof.sh wd4a "gemm_wgrad" VPF_WGROUP_DMA_TN=128 -- wstack > /dev/null 2>&1
bash tools/kprof.sh wd4b "gemm_wgrad" VPF_WGROUP_DMA_TN=0 -- wstack > /dev/null 2>&1
echo "== A/B step"; bash tools/ab.sh "VPF_WGROUP_DMA_TN=128" "VPF_WGROUP_DMA_TN=0" 3
} > gpurun_out/r05_dma5.txt 2>&1
cat gpurun_out/r05_dma5.txt | grep -v amdgpu.ids
