#!/usr/bin/env python3
"""Where does the REPLAYED hipGraph of the pre-training step run its two branches?  Device timestamps (ops.Timeline / vpf_stamp) are
captured into the graph at the branch boundaries (Pretrainer.timeline) and around Group2Emb and the two encoders; after a number of
replays the median of every mark is printed on a common time axis.  Unlike a rocprofv3 kernel trace this does not change how the
graph's nodes are dispatched, so it shows the overlap the benchmark really gets.
usage: python tools/step_timeline.py [arch=c2] [pairs=64] [replays=30]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def wrap_forward(mod, tl, fwd_begin, fwd_end, bwd_begin, pick=None):
    """Mark entry / exit of mod.forward and the arrival of the gradient at its output."""
    from vipformer_amd import ops
    orig = mod.forward

    def fwd(*a, **k):
        if fwd_begin:
            tl.mark(fwd_begin)
        out = orig(*a, **k)
        if isinstance(out, tuple):
            i = pick or 0
            return out[:i] + (ops.StampFn.apply(out[i], tl, fwd_end, bwd_begin),) + out[i + 1:]
        return ops.StampFn.apply(out, tl, fwd_end, bwd_begin)
    mod.forward = fwd


class GateFn(torch.autograd.Function):
    """Identity; its backward makes the current stream wait for `ev` (recorded somewhere in the OTHER branch's backward)."""

    @staticmethod
    def forward(ctx, x, ev, tl):
        ctx.ev, ctx.tl = ev, tl
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        torch.cuda.current_stream().wait_event(ctx.ev)
        ctx.tl.mark("img.bwd.released")
        return g, None, None


class EventFn(torch.autograd.Function):
    """Identity; its backward records `ev` on the current stream."""

    @staticmethod
    def forward(ctx, x, ev):
        ctx.ev = ev
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.ev.record(torch.cuda.current_stream())
        return g, None


def gate_image_backward(pc, im, tl, after):
    """Experiment: hold the image branch's backward until the point-cloud branch's backward reaches encoder layer `after`
    (5 .. 0 = self-attention layers, -1 = the cross-attention layer, -2 = Group2Emb)."""
    from vipformer_amd import ops
    ev = torch.cuda.Event()
    pc_ca = pc.encoder.cross_attn_1
    if after >= -1:
        def hook(ca, i):
            if ca is pc_ca and i == after:
                ev.record(torch.cuda.current_stream())
        ops.cfg.enc_bwd_hook = hook
    else:
        g2e = pc.group2emb.forward
        pc.group2emb.forward = lambda *a, **k: EventFn.apply(g2e(*a, **k), ev)
    orig = im.forward

    def fwd(*a, **k):
        out = orig(*a, **k)
        return (GateFn.apply(out[0], ev, tl),) + tuple(out[1:])
    im.forward = fwd


def main(arch="c2", pairs=64, replays=30):
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
    tl = tr.timeline = ops.Timeline(dev)
    kv_apply = ops.AdapterKVFn.apply

    def kv_marked(*a):                        # the K / V producer: the point-cloud stream's first kernels
        tl.mark("pc.kv.fwd.begin")
        out = kv_apply(*a)
        tl.mark("pc.kv.fwd.end")
        return out
    ops.AdapterKVFn.apply = kv_marked
    wrap_forward(pc.group2emb, tl, "pc.g2e.fwd.begin", "pc.g2e.fwd.end", "pc.g2e.bwd.begin")
    wrap_forward(pc.encoder, tl, "pc.enc.fwd.begin", "pc.enc.fwd.end", "pc.enc.bwd.begin")
    wrap_forward(im.encoder, tl, "img.enc.fwd.begin", "img.enc.fwd.end", "img.enc.bwd.begin")
    if os.environ.get("IMG_BWD_AFTER"):
        gate_image_backward(pc, im, tl, int(os.environ["IMG_BWD_AFTER"]))
    opt = tr.optimizer_step

    def opt_marked():
        tl.mark("adamw.begin")
        opt()
        tl.mark("adamw.end")
    tr.optimizer_step = opt_marked
    t1, t2, imgs = bench.synth_batch(pairs, A["N"], A["img"], seed=0, device=dev)
    tr.capture(t1, t2, imgs, warmup=3)
    for _ in range(5):
        tr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rows = []
    e0.record()
    for _ in range(replays):
        tr.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / replays
    for _ in range(replays):                    # (reading the marks needs a sync: timed separately above)
        for _ in range(4):                      # the marks of the LAST of four back-to-back replays: a replay launched on an idle GPU runs
            tr.replay()                         # its first ~0.5 ms at the host's enqueue rate (tools/diag_replay_host.py) and shows the
        torch.cuda.synchronize()                # second branch starting 50 - 140 us late; in steady state the host is several steps ahead
        rows.append(tl.read())
    names = list(rows[0])
    med = {n: sorted(r[n] for r in rows)[len(rows) // 2] for n in names}
    print(f"{arch}, {pairs} pairs: {ms:.3f} ms per replay back to back ({len(names)} marks in the graph)")
    for n in sorted(names, key=lambda n: med[n]):
        print(f"{med[n]:9.1f} us  {n}")
    return med


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0] if a else "c2", int(a[1]) if len(a) > 1 else 64, int(a[2]) if len(a) > 2 else 30)
