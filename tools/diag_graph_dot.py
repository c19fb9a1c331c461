#!/usr/bin/env python3
"""Dump the captured step's graph (hipGraphDebugDotPrint) and list, for the first nodes of each branch, what they depend on.
usage: tools/diag_graph_dot.py [arch] -> gpurun_out/step_graph.dot"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main(arch="c2", pairs=64):
    import bench
    from vipformer_amd import ops
    from vipformer_amd.train import Pretrainer, build_models
    begin = torch.cuda.CUDAGraph.capture_begin

    def capture_begin(self, *a, **k):
        self.enable_debug_mode()
        return begin(self, *a, **k)
    torch.cuda.CUDAGraph.capture_begin = capture_begin
    A = bench.ARCHS[arch]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    ops.rng.seed(1234)
    pc, im = build_models(**A, device=dev)
    pc.train(); im.train()
    tr = Pretrainer(pc, im)
    t1, t2, imgs = bench.synth_batch(pairs, A["N"], A["img"], seed=0, device=dev)
    tr.capture(t1, t2, imgs, warmup=3)
    os.makedirs("gpurun_out", exist_ok=True)
    path = os.path.abspath("gpurun_out/step_graph.dot")
    tr._graph.debug_dump(path)
    txt = open(path).read()
    print(len(txt), "bytes of DOT")
    edges = re.findall(r'"?([\w.]+)"?\s*->\s*"?([\w.]+)"?', txt)
    labels = dict(re.findall(r'"?([\w.]+)"?\s*\[[^\]]*label="([^"]*)"', txt))
    print(len(labels), "labelled nodes,", len(edges), "edges")
    indeg = {}
    for a, b in edges:
        indeg.setdefault(b, []).append(a)
    order = list(labels)
    for n in order[:40]:
        print(n, "|", labels[n][:70].replace("\n", " "), "| deps:", [labels.get(d, d)[:30] for d in indeg.get(n, [])])


if __name__ == "__main__":
    main(*sys.argv[1:2])
