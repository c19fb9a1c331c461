#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== tests"; timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "grouped_wgrad" 2>&1 | tail -4
echo "== c4 A/B"; bash tools/ab.sh "VPF_WGROUP_DMA=0" "VPF_WGROUP_DMA=2048" 3 --arch c4
echo "== c3 A/B"; bash tools/ab.sh "VPF_WGROUP_DMA=0" "VPF_WGROUP_DMA=2048" 2 --arch c3
echo "== ref144m4 A/B"; bash tools/ab.sh "VPF_WGROUP_DMA=0" "VPF_WGROUP_DMA=2048" 2 --arch ref144m4
} > gpurun_out/r05_c4dma.txt 2>&1
cat gpurun_out/r05_c4dma.txt | grep -v amdgpu.ids
