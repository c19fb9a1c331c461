#!/usr/bin/env python3
"""Data-parallel path on ONE GPU: two ranks (processes) share cuda:0 and exchange gradients over gloo, exercising exactly what
bench.py does at --gpus N > 1 except the RCCL transport: graph = forward + backward, then Pretrainer.exchange_and_step (region-wise
asynchronous all-reduce on the communication stream, AdamW per region).
Checks every step: the reduced gradient equals the sum of the two ranks' local gradients, parameters stay bitwise identical across
ranks, losses finite, and the parameters equal a single-process AdamW on the mean gradient.
usage: python tools/dp2_one_gpu.py [pairs=8] [steps=3]"""
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
    A = bench.ARCHS["c2"]
    torch.manual_seed(1)
    ops.rng.seed(1234 + rank)
    pc, im = build_models(**A, device=dev)
    pc.train(); im.train()
    tr = Pretrainer(pc, im, world_size=world)
    tr.broadcast_parameters(0)
    torch.manual_seed(100 + rank)
    t1, t2, imgs = bench.synth_batch(pairs, A["N"], A["img"], seed=rank, device=dev)
    tr.capture(t1, t2, imgs, warmup=2, keep_grads=True)        # (this check reads the reduced gradients after the step)
    for s in range(steps):
        tr._graph.replay()
        torch.cuda.synchronize()
        local = tr.flat.g.clone()
        p_before, m_before, v_before = tr.flat.p.clone(), tr.flat.m.clone(), tr.flat.v.clone()
        step_no = float(tr.hyper[6])
        tr.exchange_and_step()
        torch.cuda.synchronize()
        both = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        err = float((tr.flat.g - (both[0] + both[1])).abs().max())
        ps = [torch.empty_like(tr.flat.p) for _ in range(world)]
        dist.all_gather(ps, tr.flat.p)
        same = bool(torch.equal(ps[0], ps[1]))
        # single-process AdamW (torch formulas, fp32) on the MEAN gradient from the same state
        gm = (both[0] + both[1]) / world
        b1, b2, lr, eps, wd = 0.9, 0.999, 1e-3, 1e-8, 0.01
        t = step_no + 1
        m = b1 * m_before + (1 - b1) * gm
        v = b2 * v_before + (1 - b2) * gm * gm
        ref = p_before * (1 - lr * wd) - (lr / (1 - b1 ** t)) * m / (v.sqrt() / (1 - b2 ** t) ** 0.5 + eps)
        dp = float((tr.flat.p - ref).abs().max())
        losses = [float(x) for x in tr.losses]
        if rank == 0:
            print(f"step {s}: losses {losses}, |allreduce - sum of local grads|max {err:.3e}, parameters identical across ranks: {same}, "
                  f"|p - AdamW(mean grad)|max {dp:.2e}", flush=True)
        assert same and err == 0.0 and all(v == v for v in losses) and dp < 1e-5, (same, err, dp)
    dist.destroy_process_group()


def main(pairs=8, steps=3, port=29577):
    mp.spawn(worker, args=(2, pairs, steps, port), nprocs=2, join=True)
    print("dp2 on one GPU: ok")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 3)
