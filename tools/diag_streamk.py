#!/usr/bin/env python3
"""stage coverage of the stream-K grouped weight gradient: dY = 1, X[m, k] = [stage(m) == k % 128] -> dW[n, k] = 64 x (times stage k % 128
was summed into the tile), M = 8192 tokens = 128 stages in two slices"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipformer_amd import _lib, ops
M = 8192
layer = [(256, 512), (512, 256), (256, 256), (768, 256)]
stack = [(256, 512), (512, 256), (256, 256), (256, 256)] + layer * 6
_lib.debug_set("wgroup_dma", 1); _lib.debug_set("wgroup_dma_tn", 128); _lib.debug_set("wgroup_dma_sk", int(os.environ.get("SK", 1)))
st = (torch.arange(M, device="cuda") // 64)
jobs = []
for N, K in stack:
    x = (st[:, None] == (torch.arange(K, device="cuda") % 128)[None, :]).to(torch.float16)
    jobs.append((torch.ones(M, N, device="cuda", dtype=torch.float16), x, N, K, torch.zeros(N, K, device="cuda")))
wg = ops.WgradBatch(cap=ops.WgradBatch.CAP)
for dy, x, N, K, dW in jobs: wg.add(dy, x, N, K, dW, None)
wg.flush(); torch.cuda.synchronize()
def ranges(v):
    out, s = [], None
    for i, c in enumerate(v + [None]):
        if s is not None and c != v[s]: out.append(f"{s}-{i - 1}:{v[s]}"); s = None
        if s is None and c is not None: s = i
    return " ".join(out)
for i, (dy, x, N, K, dW) in enumerate(jobs):
    t = dW.view(N // 256, 256, K // 128, 128)
    uni = torch.equal(t.amax(1), t.amin(1))
    c = (t[:, 0] / 64).int()          # [ty, tx, 128]
    for ty in range(c.shape[0]):
        for tx in range(c.shape[1]):
            print(i, (N, K), "tile", ty * c.shape[1] + tx, ranges(c[ty, tx].tolist()), "" if uni else "(rows differ)")
