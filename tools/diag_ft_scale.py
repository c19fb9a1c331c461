#!/usr/bin/env python3
"""Does the part-segmentation fine-tune step (ft_partseg.py:145-176 -- NO GradScaler in the reference) lose gradient to fp16's range?
The same step twice from the same state and dropout masks: loss.backward() as the reference has it, and (loss * 2^k).backward() / 2^k.
Prints the all-parameter and the lowest per-tensor cosine between the two gradients and the share of exactly-zero gradient entries."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vipformer_amd import ops, ops_seg as S
from vipformer_amd.model.pointcloud import CrossFormer_partseg, PointCloudInputAdapter

B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 1024
torch.manual_seed(1)
ad = PointCloudInputAdapter((N, 3), 256)
m = CrossFormer_partseg(ad, 128, 256, 32, 1, 4, 8, 4, 2, 0.0, 0.1, 0.5, [2, 5, 8], 50).cuda().train()
g = torch.Generator().manual_seed(7)
pts = torch.randn(B, N, 3, generator=g)
pts = ((pts - pts.mean(1, keepdim=True)) / pts.norm(dim=2).max(dim=1)[0].view(B, 1, 1)).cuda()
onehot = torch.zeros(B, 16, device="cuda"); onehot[torch.arange(B), torch.arange(B) % 16] = 1.0
target = torch.randint(0, 50, (B, N), generator=g).cuda()
start = torch.randint(0, N, (B,), generator=g).cuda()
res = {}
for k in ("default", 0, 8, 12, 16):
    m.internal_grad_scale = k == "default"               # default: the model normalises the gradient that enters it (ops.internal_grad_scale)
    k = 0 if k == "default" else k
    tag = "model default (internal scale)" if m.internal_grad_scale else f"internal scale off, loss x 2^{k}"
    ops.rng.seed(5)
    m.zero_grad(set_to_none=True)
    real = torch.randint
    torch.randint = lambda *a, **kw: start.clone()
    try:
        with ops.rng.pinned():
            pred = m(pts, onehot)
            loss = S.cross_entropy_smooth(pred.reshape(-1, 50), target.reshape(-1), 0.2)
            (loss * float(2 ** k)).backward()
    finally:
        torch.randint = real
    key = "default" if m.internal_grad_scale else k
    res[key] = {n: p.grad.detach().double() / 2 ** k for n, p in m.named_parameters() if p.grad is not None}
    allg = torch.cat([v.flatten() for v in res[key].values()])
    print(f"{tag}: loss {float(loss):.5f}  |grad| {float(allg.norm()):.4e}  finite {bool(torch.isfinite(allg).all())}  zero entries {float((allg == 0).double().mean()):.4f}", flush=True)
ref = res[12]
for k in ("default", 0, 8, 16):
    if not all(torch.isfinite(v).all() for v in res[k].values()):
        print(f"{k} vs 2^12: non-finite gradients (overflow)"); continue
    a = torch.cat([res[k][n].flatten() for n in ref]); b = torch.cat([ref[n].flatten() for n in ref])
    per = sorted((float((res[k][n].flatten() @ ref[n].flatten()) / (res[k][n].norm() * ref[n].norm() + 1e-300)), n) for n in ref if float(ref[n].norm()) > 0)
    print(f"{k} vs loss x 2^12: all-parameter cosine {float(a @ b / (a.norm() * b.norm())):.6f}  norm ratio {float(a.norm() / b.norm()):.4f}  lowest tensors {per[:3]}")
