#!/usr/bin/env python3
"""Are the forward passes reproducible while GEMM workgroups (MFMAs fed by LDS fragment reads) share the CUs?  Both models in eval mode
(no dropout, no atomics: bitwise deterministic) on one stream, a loop of dgrad GEMMs on another; every output hashed.
Background: fps_kernel, compiled with packed-fp32 instructions (v_pk_add_f32 / v_pk_mul_f32), mis-sampled beside
such workgroups.  usage: python tools/diag_beside_gemms.py [runs=300] [pairs=8]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
H16 = torch.float16


def main(runs=300, pairs=8):
    import bench
    from vipformer_amd import ops
    from vipformer_amd.train import build_models
    A = bench.ARCHS["c2"]
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    pc, im = build_models(**A, device=dev)
    pc.eval(); im.eval()
    t1, t2, imgs = bench.synth_batch(pairs, A["N"], A["img"], seed=0, device=dev)
    pts = torch.cat([t1, t2]); imgs = imgs.permute(0, 2, 3, 1)
    start = torch.randint(0, A["N"], (2 * pairs,), device=dev)
    torch.randint = lambda *a, **k: start.clone()
    M, N, K = 12288, 512, 256
    dY = torch.randn(M, N, device=dev).to(H16); W = (torch.randn(N, K, device=dev) * 0.05).to(H16)
    side = torch.cuda.Stream()
    seen = {"pc feats": collections.Counter(), "pc backbone": collections.Counter(), "img feats": collections.Counter(), "img backbone": collections.Counter()}
    h = lambda t: hash(t.detach().float().cpu().numpy().tobytes())
    torch.cuda.synchronize()
    for r in range(runs):
        with torch.cuda.stream(side):
            for _ in range(40):
                ops.linear_dgrad(dY, W, N, K)
        with torch.no_grad():
            f, bb = pc(pts)
            fi, bbi = im(imgs)
        torch.cuda.synchronize()
        for k, t in (("pc feats", f), ("pc backbone", bb), ("img feats", fi), ("img backbone", bbi)):
            seen[k][h(t)] += 1
    for k, c in seen.items():
        print(f"   {k:13s}: {c.most_common(1)[0][1]:5d} x the usual result of {runs}, {len(c)} distinct")


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 300, int(a[1]) if len(a) > 1 else 8)
