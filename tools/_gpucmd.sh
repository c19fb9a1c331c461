#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== tests (DMA on for everything that conforms)"; VPF_WGROUP_DMA=1 timeout 1500 python3 -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -x -q -k "wgrad or stages_vs_reference_golden or models_vs_reference_golden or training_step_with_dropout or gemm" 2>&1 | tail -6
bash tools/kprof.sh wd3 "gemm_wgrad" VPF_WGROUP_DMA=1 -- wstack gemmtn > /dev/null 2>&1
echo "== A/B step"; bash tools/ab.sh "VPF_WGROUP_DMA=0" "VPF_WGROUP_DMA=2048" 3
} > gpurun_out/r05_dma4.txt 2>&1
cat gpurun_out/r05_dma4.txt | grep -v amdgpu.ids
