#!/usr/bin/env python3
"""Race detector for the captured step: with lr = wd = 0 and the dropout step pinned, every replay of the hipGraph runs the same forward
pass on the same weights, and the forward pass has no atomics -- the three losses must be BITWISE equal replay after replay.  A kernel
that reads LDS or HBM it did not write (or overruns its LDS request into a co-resident workgroup) shows up as a second loss value.
usage: python tools/diag_replay_determinism.py [arch=c2] [pairs=8] [replays=300]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main(arch="c2", pairs=8, replays=300):
    import bench
    from vipformer_amd import ops
    from vipformer_amd.train import Pretrainer, build_models
    A = bench.ARCHS[arch]
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    ops.rng.seed(1234)
    pc, im = build_models(**A, device=dev)
    pc.train(); im.train()
    tr = Pretrainer(pc, im)
    tr.hyper[0] = 0.0; tr.hyper[4] = 0.0                      # lr = wd = 0: the weights stay what they are
    t1, t2, imgs = bench.synth_batch(pairs, A["N"], A["img"], seed=0, device=dev)
    start = torch.randint(0, A["N"], (2 * pairs,), device=dev)
    real = torch.randint
    torch.randint = lambda *a, **k: start.clone()             # farthest_point_sample's start indices (utils.py:71): the same every replay
    try:
        tr.capture(t1, t2, imgs, warmup=2, keep_grads=True)
    finally:
        torch.randint = real
    seen = collections.Counter()
    gsum = collections.Counter()
    for i in range(replays):
        ops.rng.state("cuda")[2] = 0                          # the same dropout masks every replay
        losses = tr.replay()
        torch.cuda.synchronize()
        seen[tuple(float(x) for x in losses)] += 1
        gsum[round(float(tr.flat.g.double().abs().sum()), 3)] += 1
    print(f"{arch}, {pairs} pairs, {replays} replays: {len(seen)} distinct loss triples")
    for k, v in seen.most_common(8):
        print(f"   {v:5d} x {k}")
    print(f"   |gradient| sums (fp32 atomics: last digits vary): {len(gsum)} distinct, range {min(gsum)} .. {max(gsum)}")
    return len(seen)


if __name__ == "__main__":
    a = sys.argv[1:]
    n = main(a[0] if a else "c2", int(a[1]) if len(a) > 1 else 8, int(a[2]) if len(a) > 2 else 300)
    sys.exit(0 if n == 1 else 1)
