#!/usr/bin/env python3
"""Which stage stops being reproducible when ANOTHER process keeps the GPU busy?  (Two ranks sharing one GPU -- tools/dp2_one_gpu.py --
showed a few percent of corrupted steps; a GPU of its own never did.)
  python tools/diag_shared_gpu.py noise [seconds]     replay the captured step in a loop (the neighbour)
  python tools/diag_shared_gpu.py check [runs]        forward + backward of both models, eagerly, `runs` times on the same inputs with the
                                                       same dropout masks; per stage: how many runs differ from the first one (forward
                                                       stages bitwise, gradients beyond fp32-atomic noise)"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def build(pairs):
    import bench
    from vipformer_amd import ops
    from vipformer_amd.train import Pretrainer, build_models
    A = bench.ARCHS["c2"]
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    ops.rng.seed(1234)
    pc, im = build_models(**A, device=dev)
    pc.train(); im.train()
    tr = Pretrainer(pc, im)
    tr.hyper[0] = 0.0; tr.hyper[4] = 0.0
    t1, t2, imgs = bench.synth_batch(pairs, A["N"], A["img"], seed=0, device=dev)
    return A, dev, pc, im, tr, t1, t2, imgs


def noise(seconds):
    A, dev, pc, im, tr, t1, t2, imgs = build(64)
    tr.capture(t1, t2, imgs, warmup=2)
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            tr.replay()
        torch.cuda.synchronize()
        n += 20
    print(f"noise: {n} replays", flush=True)


def check(runs):
    from vipformer_amd import ops
    from vipformer_amd.model.pointcloud import utils as U
    A, dev, pc, im, tr, t1, t2, imgs = build(8)
    start = torch.randint(0, A["N"], (16,), device=dev)
    stages = collections.OrderedDict()

    def keep(name, t):
        stages[name] = t.detach().clone()
        return t
    real_dp = U.divide_patches

    def dp(*a, **k):
        nb, ct = real_dp(*a, **k)
        keep("fwd 0 centres", ct); keep("fwd 1 groups", nb)
        return nb, ct
    U.divide_patches = dp
    g2e = pc.group2emb.forward
    pc.group2emb.forward = lambda *a, **k: keep("fwd 2 group2emb", g2e(*a, **k))
    enc = pc.encoder.forward

    def encf(x, pos, kv, *a, **k):
        keep("fwd 3 pos", pos); keep("fwd 4 kv (adapter)", kv)
        return keep("fwd 5 pc encoder", enc(x, pos, kv, *a, **k))
    pc.encoder.forward = encf
    ienc = im.encoder.forward

    def iencf(x, pos, kv, *a, **k):
        keep("fwd 6 img patches", x)
        return keep("fwd 7 img encoder", ienc(x, pos, kv, *a, **k))
    im.encoder.forward = iencf
    real = torch.randint
    torch.randint = lambda *a, **k: start.clone()
    first, bad = None, collections.Counter()
    names = None
    for r in range(runs):
        ops.rng.state("cuda")[2] = 0
        with ops.rng.pinned():
            losses = tr.forward_backward(t1, t2, imgs)
        torch.cuda.synchronize()
        cur = collections.OrderedDict(stages)
        cur["fwd 8 losses"] = torch.stack([x.detach().float().reshape(()) for x in losses])
        groups = collections.OrderedDict()
        for m, tag in ((pc, "pc"), (im, "img")):
            for k, p in m.named_parameters():
                parts = k.split(".")
                top = ".".join(parts[:3]) if parts[0] == "encoder" else parts[0]
                groups.setdefault(f"bwd {tag}.{top}", []).append(p.grad.reshape(-1))
        for k, v in groups.items():
            cur[k] = torch.cat(v).clone()
        if first is None:
            first, names = {k: v.clone() for k, v in cur.items()}, list(cur)
            continue
        for k in names:
            a, b = first[k], cur[k]
            if k.startswith("fwd"):
                same = torch.equal(a, b)
            else:
                same = float((a.double() - b.double()).norm() / (a.double().norm() + 1e-30)) < 1e-4
            bad[k] += 0 if same else 1
    torch.randint = real
    print(f"{runs} runs; stages that differed from the first run:")
    for k in names:
        print(f"   {bad[k]:5d}  {k}")


if __name__ == "__main__":
    if sys.argv[1] == "noise":
        noise(float(sys.argv[2]) if len(sys.argv) > 2 else 60)
    else:
        check(int(sys.argv[2]) if len(sys.argv) > 2 else 200)
