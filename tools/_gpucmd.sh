#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for r in 50 65 80; do
  echo "== ramp $r"; VPF_WGROUP_DMA_RAMP=$r timeout 600 python3 tools/microbench.py gemmtn 2>&1 | grep "dW3\[:,128\|kv dW" | cut -c1-90
done
echo "== A/B step ramp 50 vs 80"; bash tools/ab.sh "VPF_WGROUP_DMA_RAMP=50" "VPF_WGROUP_DMA_RAMP=80" 3
} > gpurun_out/r05_ramp2.txt 2>&1
cat gpurun_out/r05_ramp2.txt | grep -v amdgpu.ids
