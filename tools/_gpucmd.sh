#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp VPF_SCRATCH=/tmp/vpf_prof
mkdir -p $VPF_SCRATCH
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch
from vipformer_amd import _lib, ops
M=12288
layer = [(256, 512), (512, 256), (256, 256), (768, 256)]
stack = [(256, 512), (512, 256), (256, 256), (256, 256)] + layer * 6
g = torch.Generator().manual_seed(7)
ops_in = [(torch.randn(M, N, generator=g).cuda().half(), torch.randn(M, K, generator=g).cuda().half(), N, K) for N, K in stack]
_lib.debug_set("wgroup_dma", 1)
for tn in (128, 32, 33):
    _lib.debug_set("wgroup_dma_tn", tn)
    outs = [(torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")) for _, _, N, K in ops_in]
    wg = ops.WgradBatch(cap=ops.WgradBatch.CAP)
    for (dy, x, N, K), (dW, db) in zip(ops_in, outs): wg.add(dy, x, N, K, dW, db)
    wg.flush(); torch.cuda.synchronize()
    worst = 0.0
    for (dy, x, N, K), (dW, db) in zip(ops_in, outs):
        ref = dy.float().t() @ x.float()
        worst = max(worst, ((dW - ref).norm() / ref.norm()).item(), ((db - dy.float().sum(0)).norm() / dy.float().sum(0).norm()).item())
    print("tn", tn, "worst rel", worst)
PY
for tn in 128 32 33; do echo "== kprof wstack tn=$tn"; bash tools/kprof.sh w$tn wgrad_dma VPF_WGROUP_DMA_TN=$tn -- wstack; done
echo "== A/B step"; bash tools/ab.sh "VPF_WGROUP_DMA_TN=128" "VPF_WGROUP_DMA_TN=32" 3
