#!/usr/bin/env python3
"""Per-stage error budget of the bf16 rounding points -- oracle against oracle, CPU only.

    python -m tests.rounding_budget [arch=c1] [pairs=4] [bf16|fp16|fp16s|fp16s1] [quick] [fixture] > profiles/r04_rounding_budget_<fwd>_<arch>.txt

`fixture` (round 6): the conditions of tests/golden/fullsize_<arch>.npz -- inputs of seeds 930 / 931, every dropout at 0 -- and, beside
the exact cosines, the cosine over the strided samples of helpers.grad_sample that the fixture keeps: the per-configuration NT-Xent
floors of tests/test_fullsize_gpu.py FIXTURE_FLOORS come from these tables (profiles/r06_rounding_budget_fixture_*.txt).

The fp32 oracle (pinned against the reference) is the baseline.  Each rounding point of the HIP path (oracle/torch_oracle.py:
FWD_TAGS -- activations / weights stored as bf16 -- and BWD_TAGS -- gradients stored as bf16 operands of the backward products) is
switched on ALONE, then all of them together, in train mode with the real dropout probabilities and ONE fixed set of masks; for
every variant the table gives
  * backbone rel-L2 and NT-Xent loss difference against fp32,
  * the all-parameter / lowest per-tensor gradient cosine against fp32 for a loss LINEAR in the backbone features (what the HIP
    backward kernels are held to) and for the pre-training loss (NT-Xent on the BatchNorm head at temperature 0.1).
It answers which rounding costs the gradient its angle, and how much of the HIP-vs-oracle differences the tests see is the
precision of the data path rather than kernel logic.
"""
import sys

import numpy as np
import torch

from oracle import torch_oracle as O
from tests import helpers as Hh


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


DROPS = (0.1, 0.5)


def run(name, B, tags, backward, pts, start, imgs, masks, Rb, Rbi):
    a = Hh.ARCHS[name]
    arch = O.Arch(D=a["D"], H=a["H"], G=a["G"], K=a["K"], S=a["S"], MR=a["MR"], img=a["img"], patch=a["patch"], atten_drop=DROPS[0], mlp_drop=DROPS[1])
    pc_sd = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pc_{name}.json"), 100)
    im_sd = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_img_{name}.json"), 200)
    isparam = lambda k, v: v.dtype == torch.float32 and "running" not in k and "cross_attn_1." not in k
    pcp = {k: v.clone().requires_grad_() for k, v in pc_sd.items() if isparam(k, v)}
    imp = {k: v.clone().requires_grad_() for k, v in im_sd.items() if isparam(k, v)}
    s1 = dict(pc_sd); s1.update(pcp); s2 = dict(im_sd); s2.update(imp)
    for s in (s1, s2):
        for k in list(s):
            if "cross_attn_1." in k:
                s[k] = s[k.replace("cross_attn_1.", "cross_attn_n.")]
    with EMU(only=tags, backward=backward):
        f, bb = O.pc_forward(s1, pts, start, arch, True, masks[0], {})
        fi, bbi = O.img_forward(s2, imgs, arch, True, masks[1], {})
        loss = O.ntxent(f[:B], f[B:]) + O.ntxent((f[:B] + f[B:]) / 2, fi)
        ((bb * Rb).sum() + (bbi * Rbi).sum()).backward(retain_graph=True)
        lin = {k: v.grad.clone() for d in (pcp, imp) for k, v in d.items() if v.grad is not None}
        for d in (pcp, imp):
            for v in d.values():
                v.grad = None
        loss.backward()
        ntx = {k: v.grad.clone() for d in (pcp, imp) for k, v in d.items() if v.grad is not None}
    return dict(bb=bb.detach(), bbi=bbi.detach(), loss=float(loss), lin=lin, ntx=ntx)


EMU = O.emulate_bf16


def main():
    global EMU
    name = sys.argv[1] if len(sys.argv) > 1 else "c1"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    fwd = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    import functools
    EMU = {"bf16": O.emulate_bf16, "fp16": O.emulate_fp16, "fp16s": functools.partial(O.emulate_fp16, grad_scale=65536.0),
           "fp16s1": functools.partial(O.emulate_fp16, grad_scale=1.0)}[fwd]
    quick = "quick" in sys.argv[4:]                             # only the combined rows
    fixture = "fixture" in sys.argv[4:]
    global DROPS
    if fixture:
        DROPS = (0.0, 0.0)
    a = Hh.ARCHS[name]
    torch.manual_seed(0)
    s_pts, s_img = (930, 931) if fixture else (300, 400)
    pts = Hh.synth_points(s_pts, 2 * B, a["N"]); start = Hh.synth_start(s_pts, 2 * B, a["N"])
    imgs = Hh.synth_images(s_img, B, a["img"], a["img"])
    T = (a["img"] // a["patch"]) ** 2

    class FixedMasks(O.Masks):        # torch Bernoulli masks drawn ONCE per site and replayed for every variant
        def __init__(self, seed):
            super().__init__("given", {})
            self.g = torch.Generator().manual_seed(seed)

        def apply(self, x, site, p):
            if p <= 0.0:
                return x
            if site not in self.table:
                self.table[site] = (torch.rand(x.shape, generator=self.g) >= p)
            return super().apply(x, site, p)

    masks = (FixedMasks(1), FixedMasks(2))
    Rb, Rbi = Hh.synth_like(700, (2 * B, 2 * a["D"])), Hh.synth_like(701, (B, 2 * a["D"]))
    ref = run(name, B, (), False, pts, start, imgs, masks, Rb, Rbi)
    variants = [] if quick else [(t, (t,), t in O.BWD_TAGS) for t in O.FWD_TAGS + O.BWD_TAGS]
    variants += [("all forward", O.FWD_TAGS, False), ("all backward", O.BWD_TAGS, True), ("all (the HIP data path)", O.FWD_TAGS + O.BWD_TAGS, True)]
    print(f"# forward operands rounded to {fwd[:4]} (gradient operands: " + {"fp16s": "fp16 at loss scale 65536", "fp16s1": "fp16, no loss scale"}.get(fwd, "bf16") + ")")
    print(f"# rounding budget, arch {name}, {B} pairs, train mode, dropout {DROPS[0]} / {DROPS[1]}" + (" (the fullsize fixture's inputs)" if fixture else " with fixed masks")
          + f"; baseline = fp32 oracle (loss {ref['loss']:.5f})")
    print(f"{'rounding point(s)':28s} {'pc bb rel':>10s} {'img bb rel':>10s} {'dloss':>9s} | linear loss: {'all cos':>9s} {'lowest':>8s} | NT-Xent: {'all cos':>9s} {'median':>8s} {'lowest':>8s}"
          + (" | strided samples: lin all cos, NT-Xent all cos" if fixture else ""))
    for label, tags, bwd in variants:
        r = run(name, B, tags, bwd, pts, start, imgs, masks, Rb, Rbi)
        rel = lambda x, y: float((x - y).double().norm() / y.double().norm())
        out = []
        for key in ("lin", "ntx"):
            ks = [k for k in ref[key] if k in r[key] and not k.endswith(("first_conv.0.bias", "first_conv.3.bias", "second_conv.0.bias"))]
            allc = cosine(torch.cat([r[key][k].flatten() for k in ks]), torch.cat([ref[key][k].flatten() for k in ks]))
            per = sorted(cosine(r[key][k], ref[key][k]) for k in ks)
            smp = cosine(torch.cat([Hh.grad_sample(r[key][k]) for k in ks]), torch.cat([Hh.grad_sample(ref[key][k]) for k in ks]))
            # per tensor over the samples, as the fixture test forms it: tensors with >= 64 samples and >= 0.1 % of the largest norm
            nmax = max(float(ref[key][k].double().norm()) for k in ks)
            sper = sorted(cosine(Hh.grad_sample(r[key][k]), Hh.grad_sample(ref[key][k])) for k in ks
                          if ref[key][k].numel() >= 64 and float(ref[key][k].double().norm()) > 1e-3 * nmax)
            heavy = sorted(cosine(Hh.grad_sample(r[key][k]), Hh.grad_sample(ref[key][k])) for k in ks
                           if ref[key][k].numel() >= 64 and float(ref[key][k].double().norm()) >= 1e-2 * nmax)
            out.append((allc, float(np.median(per)), per[0], smp, float(np.median(sper)), heavy[0]))
        print(f"{label:28s} {rel(r['bb'], ref['bb']):10.2e} {rel(r['bbi'], ref['bbi']):10.2e} {abs(r['loss'] - ref['loss']):9.2e} | "
              f"{'':13s}{out[0][0]:9.5f} {out[0][2]:8.5f} | {'':9s}{out[1][0]:9.5f} {out[1][1]:8.5f} {out[1][2]:8.5f}"
              + (f" | {out[0][3]:9.5f} {out[1][3]:9.5f} | per-tensor median over samples: {out[0][4]:8.5f} {out[1][4]:8.5f} | lowest among tensors with >= 1 % of the largest norm: {out[0][5]:8.5f} {out[1][5]:8.5f}" if fixture else ""), flush=True)


if __name__ == "__main__":
    main()
