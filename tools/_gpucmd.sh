#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp VPF_SCRATCH=/tmp/vpf_prof
mkdir -p $VPF_SCRATCH
SK=1 python3 tools/diag_streamk.py 2>&1 | grep -v "0-127:1 *$" | grep -v amdgpu.ids | head
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "grouped_wgrad" --tb=line 2>&1 | grep -v "^E   *+\|tensor" | cut -c1-400 | tail -8
echo "== kprof wstack sk=0"; bash tools/kprof.sh w0 wgrad_dma VPF_WGROUP_DMA_SK=0 -- wstack
echo "== kprof wstack sk=1"; bash tools/kprof.sh w1 wgrad_dma VPF_WGROUP_DMA_SK=1 -- wstack
echo "== A/B step"; bash tools/ab.sh "VPF_WGROUP_DMA_SK=0" "VPF_WGROUP_DMA_SK=1" 4
