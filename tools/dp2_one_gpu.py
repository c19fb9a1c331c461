#!/usr/bin/env python3
"""Data-parallel path on ONE GPU: two ranks (processes) share cuda:0 and exchange gradients over gloo, exercising exactly what
bench.py does at --gpus N > 1 except the RCCL transport: graph = forward + backward, eager all-reduce of the flat gradient, AdamW.
Checks: parameters stay identical across ranks after every step, losses finite, and the averaged gradient equals the mean of
the two ranks' local gradients.   usage: python tools/dp2_one_gpu.py [pairs=8] [steps=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, pairs, steps, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from vipformer_amd import ops
    from vipformer_amd.train import Pretrainer, build_models
    torch.manual_seed(1)
    ops.rng.seed(1234 + rank)
    pc, im = build_models(**bench.ARCH, device=dev)
    pc.train(); im.train()
    tr = Pretrainer(pc, im, world_size=world)
    tr.broadcast_parameters(0)
    torch.manual_seed(100 + rank)
    t1, t2, imgs = bench.synth_batch(pairs, bench.ARCH["N"], bench.ARCH["img"], seed=rank, device=dev)
    tr.capture(t1, t2, imgs, warmup=2)
    for s in range(steps):
        tr._graph.replay()
        torch.cuda.synchronize()
        local = tr.flat.g.clone()
        tr.allreduce_gradients()
        both = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        err = float((tr.flat.g - (both[0] + both[1])).abs().max())
        tr.optimizer_step()
        torch.cuda.synchronize()
        ps = [torch.empty_like(tr.flat.p) for _ in range(world)]
        dist.all_gather(ps, tr.flat.p)
        same = bool(torch.equal(ps[0], ps[1]))
        losses = [float(x) for x in tr.losses]
        if rank == 0:
            print(f"step {s}: losses {losses}, |allreduce - sum of local grads|max {err:.3e}, parameters identical across ranks: {same}", flush=True)
        assert same and err == 0.0 and all(v == v for v in losses)
    dist.destroy_process_group()


if __name__ == "__main__":
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    mp.spawn(worker, args=(2, pairs, steps, 29577), nprocs=2, join=True)
    print("dp2 on one GPU: ok")
