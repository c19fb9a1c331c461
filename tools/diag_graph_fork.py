#!/usr/bin/env python3
"""When does the SECOND branch of a replayed two-stream hipGraph start?  main: mark, then `n_main` launches of ~`us` microseconds each
(a square matmul sized for it); side (forked before them): mark, then its own chain; join; mark.  Device stamps (ops.Timeline) of a
replayed graph, medians over replays.   usage: tools/diag_graph_fork.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vipformer_amd import ops


def run(n_main, n_side, dim_main, dim_side, side_first, replays=20):
    dev = torch.device("cuda", 0)
    a = torch.randn(dim_main, dim_main, device=dev, dtype=torch.float16)
    b = torch.randn(dim_side, dim_side, device=dev, dtype=torch.float16)
    tl = ops.Timeline(dev)
    side = torch.cuda.Stream()
    cap = torch.cuda.Stream()

    def body():
        main = torch.cuda.current_stream()
        tl.mark("begin")
        side.wait_stream(main)

        def side_part():
            with torch.cuda.stream(side):
                tl.mark("side.first")
                x = b
                for _ in range(n_side):
                    x = x @ b
                tl.mark("side.last")

        def main_part():
            y = a
            for i in range(n_main):
                y = y @ a
                if i == 0:
                    tl.mark("main.after1")
            tl.mark("main.last")
        if side_first:
            side_part(); main_part()
        else:
            main_part(); side_part()
        main.wait_stream(side)
        tl.mark("end")
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        for _ in range(3):
            body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        body()
    for _ in range(5):
        g.replay()
    rows = []
    for _ in range(replays):
        g.replay(); g.replay(); g.replay()
        torch.cuda.synchronize()
        rows.append(tl.read())
    med = {n: sorted(r[n] for r in rows)[len(rows) // 2] for n in rows[0]}
    print(f"main {n_main} x matmul({dim_main}), side {n_side} x matmul({dim_side}), side captured {'first' if side_first else 'last'}: "
          + "  ".join(f"{n} {med[n]:.0f}" for n in sorted(med, key=lambda n: med[n])))


if __name__ == "__main__":
    for side_first in (True, False):
        run(8, 8, 2048, 2048, side_first)
        run(8, 8, 4096, 1024, side_first)
        run(40, 8, 1024, 2048, side_first)
    run(1, 1, 8192, 1024, True)
