#!/usr/bin/env python3
"""How long does the HOST spend in one replay of the captured step, and does a second instantiated graph (alternating replays) change the
step time?  If hipGraphLaunch of an executable graph waits for that graph's previous launch, the host can only start enqueueing step
k + 1 when step k has finished, and the start of every step runs at the host's enqueue rate.   usage: tools/diag_replay_host.py [arch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main(arch="c2", pairs=64):
    import bench
    from vipformer_amd import ops
    from vipformer_amd.train import Pretrainer, build_models
    A = bench.ARCHS[arch]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    ops.rng.seed(1234)
    pc, im = build_models(**A, device=dev)
    pc.train(); im.train()
    tr = Pretrainer(pc, im)
    t1, t2, imgs = bench.synth_batch(pairs, A["N"], A["img"], seed=0, device=dev)
    tr.capture(t1, t2, imgs, warmup=3)
    for _ in range(10):
        tr.replay()
    torch.cuda.synchronize()
    for label, n in (("back to back", 40),):
        host = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            t = time.perf_counter()
            tr.replay()
            host.append((time.perf_counter() - t) * 1e3)
        e1.record()
        torch.cuda.synchronize()
        host.sort()
        print(f"{label}: {e0.elapsed_time(e1) / n:.3f} ms per step on the device; host time in replay(): median {host[len(host) // 2]:.3f} ms, "
              f"min {host[0]:.3f}, max {host[-1]:.3f}")
    # an idle GPU: how long does the enqueue alone take?
    torch.cuda.synchronize()
    t = time.perf_counter(); tr.replay(); dt = (time.perf_counter() - t) * 1e3
    torch.cuda.synchronize()
    print(f"enqueue of one replay on an idle GPU: {dt:.3f} ms of host time")


if __name__ == "__main__":
    main(*sys.argv[1:2])
