#!/usr/bin/env python3
"""Does the reference's part-segmentation fine-tune step (ft_partseg.py:145-176 on the mirrored CrossFormer_partseg) capture into a
hipGraph as it stands, and what does a replay cost next to the eager step?   usage: tools/diag_c5_graph.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vipformer_amd import ops_seg as S
from vipformer_amd.model.pointcloud import CrossFormer_partseg, PointCloudInputAdapter

dev = torch.device("cuda", 0)
a = bench.ARCHS["c3"]
B, N = 16, 1024
torch.manual_seed(1)
ad = PointCloudInputAdapter((N, 3), a["D"])
m = CrossFormer_partseg(ad, a["G"], a["D"], a["K"], 1, a["H"], a["S"], a["H"], a["MR"], 0.0, 0.1, 0.5, [2, 5, 8], 50).to(dev)
m.train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-3, capturable=True)
g = torch.Generator(device="cpu").manual_seed(7)
pts = torch.randn(B, N, 3, generator=g)
pts = ((pts - pts.mean(1, keepdim=True)) / pts.norm(dim=2).max(dim=1)[0].view(B, 1, 1)).to(dev)
onehot = torch.zeros(B, 16, device=dev); onehot[torch.arange(B), torch.arange(B) % 16] = 1.0
target = torch.randint(0, 50, (B, N), generator=g).to(dev)
out = {}


def step():
    opt.zero_grad(set_to_none=True)
    pred = m(pts, onehot)
    loss = S.cross_entropy_smooth(pred.reshape(-1, 50), target.reshape(-1), 0.2)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 10, norm_type=2)
    opt.step()
    out["loss"] = loss


def timed(fn, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(5):
        step()
torch.cuda.current_stream().wait_stream(s)
print(f"eager: {timed(step):.3f} ms/step, loss {float(out['loss']):.4f}")
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=s):
    step()
for _ in range(3):
    gr.replay()
print(f"captured: {timed(gr.replay):.3f} ms/step, loss {float(out['loss']):.4f}")
l0 = float(out["loss"])
for _ in range(50):
    gr.replay()
print(f"loss after 50 more replays: {float(out['loss']):.4f} (was {l0:.4f})")
