"""GPU parity at BASELINE.json's FULL batch sizes: the goldens of test_modules_gpu.py hold 2-4 pairs; here
the HIP path runs configs 2, 3 and 4 at their per-GPU batch (64 / 32 / 16 pairs) against the oracle run live on the host
(the oracle itself is pinned against the reference for all three architectures by tests/test_oracle_golden.py).

  * eval-mode forward of both models: backbone and projected features, rel-L2 <= 2e-2 (a whole network of bf16 operand rounding,
    the bound test_models_vs_reference_golden uses);
  * c2 at 32 pairs, train mode with the real dropout probabilities and the kernels' own masks: the NT-Xent loss and its gradients
    with the projection head's BatchNorm over 64 / 32 samples, where the contract's bounds (loss abs <= 5e-3, gradient cosine)
    are meaningful -- the 4-pair fixtures normalise over 8 samples and amplify every forward difference.
"""
import numpy as np
import pytest
import torch

from tests import helpers as Hh
from tests.test_modules_gpu import Checks, _site_masks, build, cosine, forced_start, rel, report

pytestmark = pytest.mark.gpu

FULL = {"c1": 64, "c3": 32, "c4": 16}          # pairs per GPU of BASELINE configs[1], [2], [3]


def _oracle_sd(name):
    pc_sd = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pc_{name}.json"), 100)
    im_sd = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_img_{name}.json"), 200)
    return pc_sd, im_sd


@pytest.mark.parametrize("name", ["c1", "c3", "c4"])
def test_full_batch_eval_forward_vs_oracle(name):
    from oracle import torch_oracle as O
    pc, im, a = build(name)
    B = FULL[name]
    ck = Checks(f"fullsize-eval[{name}, {B} pairs]")
    pts = Hh.synth_points(900, 2 * B, a["N"]); start = Hh.synth_start(900, 2 * B, a["N"])
    imgs = Hh.synth_images(901, B, a["img"], a["img"])
    pc.eval(); im.eval()
    with torch.no_grad(), forced_start(start.cuda()):
        f, bb = pc(pts.cuda())
        fi, bbi = im(imgs.cuda())
    arch = O.Arch(D=a["D"], H=a["H"], G=a["G"], K=a["K"], S=a["S"], MR=a["MR"], img=a["img"], patch=a["patch"])
    pc_sd, im_sd = _oracle_sd(name)
    with torch.no_grad():
        fr, bbr = O.pc_forward(pc_sd, pts, start, arch, False)
        fir, bbir = O.img_forward(im_sd, imgs, arch, False)
    ck.lt("pc backbone rel", rel(bb, bbr), 2e-2)
    ck.lt("pc feats rel", rel(f, fr), 2e-2)
    ck.lt("img backbone rel", rel(bbi, bbir), 2e-2)
    ck.lt("img feats rel", rel(fi, fir), 2e-2)
    # per-sample worst case (a single bad cloud must not hide in the batch norm of the error)
    per = ((bb.cpu().double() - bbr.double()).norm(dim=1) / bbr.double().norm(dim=1)).max().item()
    ck.lt("pc backbone worst-sample rel", per, 4e-2)
    ck.done()


def test_ntxent_loss_and_gradients_at_32_pairs():
    """c2 architecture, 32 pairs (BatchNorm of the projection head over 64 clouds / 32 images), dropout 0.1 / 0.5 with exported
    masks.  Bounds: loss abs <= 5e-3 vs the fp32 oracle (measured 3e-5); NT-Xent gradients vs the bf16-emulating
    oracle: all-parameter cosine >= 0.985, median per-tensor >= 0.98, lowest per-tensor >= 0.965 (measured 0.9924 / 0.9911 /
    0.9856 -- against 0.65 - 0.85 for the 4-pair fixtures, whose head BatchNorm normalises over 8 samples -- and 0.9892 / 0.9897 /
    0.9788 for the SAME kernels compiled without packed-fp32 instructions: a different last bit here and there moves these three
    numbers by 0.003 - 0.007, which is the band the floors leave).  What is left is the
    temperature: the projected features agree to 1.6e-2 (bf16 through 7 layers with p = 0.5 dropout scaling), the logits are
    features / 0.1, so dL/dfeats turns by ~1e-1 in angle; every parameter's gradient inherits that one rotation, which is why the
    per-tensor cosines sit in a narrow band (0.986 - 0.995) instead of a few outliers.  The backward KERNELS are checked to
    >= 0.999 by the linear-loss comparisons of test_modules_gpu.py."""
    from oracle import torch_oracle as O
    from vipformer_amd import ops
    name, B = "c1", 32
    ops.rng.seed(4321)
    with ops.rng.pinned():
        pc, im, a = build(name, (0.1, 0.5))
        ck = Checks(f"ntxent-32[{name}]")
        pts = Hh.synth_points(910, 2 * B, a["N"]); start = Hh.synth_start(910, 2 * B, a["N"])
        imgs = Hh.synth_images(911, B, a["img"], a["img"])
        pc.train(); im.train(); pc.zero_grad(); im.zero_grad()
        with forced_start(start.cuda()):
            f, bb = pc(pts.cuda())
        fi, bbi = im(imgs.cuda())
        f1, f2 = f[:B], f[B:]
        loss = ops.ntxent_loss(f1, f2, 0.1) + ops.ntxent_loss((f1 + f2) / 2, fi, 0.1)
        loss.backward()
        arch = O.Arch(D=a["D"], H=a["H"], G=a["G"], K=a["K"], S=a["S"], MR=a["MR"], img=a["img"], patch=a["patch"],
                      atten_drop=0.1, mlp_drop=0.5)
        pm = O.Masks("given", _site_masks(pc, (2 * B, a["G"]), a["N"], a, "cuda"))
        T = (a["img"] // a["patch"]) ** 2
        imk = O.Masks("given", _site_masks(im, (B, T), T, a, "cuda"))
    pc_sd, im_sd = _oracle_sd(name)
    isparam = lambda k, v: v.dtype == torch.float32 and "running" not in k and "cross_attn_1." not in k
    pcp = {k: v.clone().requires_grad_() for k, v in pc_sd.items() if isparam(k, v)}
    imp = {k: v.clone().requires_grad_() for k, v in im_sd.items() if isparam(k, v)}
    s1 = dict(pc_sd); s1.update(pcp); s2 = dict(im_sd); s2.update(imp)
    for s in (s1, s2):
        for k in list(s):
            if "cross_attn_1." in k:
                s[k] = s[k.replace("cross_attn_1.", "cross_attn_n.")]
    with torch.no_grad():
        fr, bbr = O.pc_forward(s1, pts, start, arch, True, pm, {})
        fir, bbir = O.img_forward(s2, imgs, arch, True, imk, {})
        lref = O.ntxent(fr[:B], fr[B:]) + O.ntxent((fr[:B] + fr[B:]) / 2, fir)
    report(f"ntxent-32 loss hip {loss.item():.5f} fp32 oracle {lref.item():.5f}")
    ck.lt("pc feats rel (fp32 oracle)", rel(f, fr), 2e-2)
    ck.lt("img feats rel (fp32 oracle)", rel(fi, fir), 2e-2)
    ck.lt("loss abs diff vs fp32 oracle (|loss| ~ 9)", abs(loss.item() - lref.item()), 5e-3)
    with O.emulate_bf16():
        fe, bbe = O.pc_forward(s1, pts, start, arch, True, pm, {})
        fie, bbie = O.img_forward(s2, imgs, arch, True, imk, {})
        le = O.ntxent(fe[:B], fe[B:]) + O.ntxent((fe[:B] + fe[B:]) / 2, fie)
    le.backward()
    ck.lt("[emulated] loss abs diff", abs(loss.item() - le.item()), 5e-3)
    cosines = []
    for model, ref in ((pc, pcp), (im, imp)):
        for k, p in model.named_parameters():
            r = ref[k].grad
            if r is None or p.grad is None or k.endswith(("first_conv.0.bias", "first_conv.3.bias", "second_conv.0.bias")):
                continue
            cosines.append((cosine(p.grad, r), k, float(r.norm())))
    cosines.sort()
    for cc, k, nr in cosines[:6]:
        report(f"ntxent-32 lowest grad cosine {cc:.5f} {k} |ref| {nr:.2e}")
    hip = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).cpu() for m in (pc, im) for _, p in m.named_parameters()])
    refg = torch.cat([(r[k].grad if r[k].grad is not None else torch.zeros_like(r[k])).reshape(-1)
                      for m, r in ((pc, pcp), (im, imp)) for k, _ in m.named_parameters()])
    ck.gt("[emulated] NT-Xent all-parameter gradient cosine", cosine(hip, refg), 0.985)
    ck.gt("[emulated] NT-Xent median per-tensor gradient cosine", float(np.median([c[0] for c in cosines])), 0.98)
    ck.gt("[emulated] NT-Xent lowest per-tensor gradient cosine", cosines[0][0], 0.965)
    ck.done()
