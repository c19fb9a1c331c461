"""GPU parity at BASELINE.json's FULL batch sizes: the goldens of test_modules_gpu.py hold 2-4 pairs; here
the HIP path runs configs 2, 3 and 4 at their per-GPU batch (64 / 32 / 16 pairs) against the oracle run live on the host
(the oracle itself is pinned against the reference for all three architectures by tests/test_oracle_golden.py).

  * eval-mode forward of both models: backbone and projected features, rel-L2 <= 2e-3 (SURVEY 8c's bound for fp16 operands);
  * the c2 architecture at 32 pairs, config 3 at 32 and config 4 at 16 pairs, the reference scripts' own geometry (2048 points,
    144 x 144 / patch 12) at 16, train mode with the real dropout probabilities and the kernels' own masks: the NT-Xent loss
    (abs <= 5e-3, the contract's bound), features behind the BatchNorm head <= 1e-2, and the gradients of every parameter for a
    linear and for the pre-training loss against the fp32 oracle with constant floors (test_modules_gpu.FLOORS);
  * the c2 train step at the BENCHMARKED batch, 64 pairs (round 5): the same checks as the 32-pair case with the
    oracle's backward passes cut to the pre-training loss (test_c2_train_step_at_the_benchmarked_batch_64_pairs).
"""
import numpy as np
import pytest
import torch

from tests import helpers as Hh
from tests.test_modules_gpu import Checks, _site_masks, build, cosine, forced_start, rel, report

pytestmark = pytest.mark.gpu

FULL = {"c1": 64, "c3": 32, "c4": 16, "ref144": 64}          # pairs per GPU of BASELINE configs[1], [2], [3]; the reference scripts' geometry


def _oracle_sd(name):
    pc_sd = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pc_{name}.json"), 100)
    im_sd = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_img_{name}.json"), 200)
    return pc_sd, im_sd


@pytest.mark.parametrize("name", ["c1", "c3", "c4", "ref144"])
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
    ck.lt("pc backbone rel", rel(bb, bbr), 2e-3)           # SURVEY 8c, fp16 operands: forward activations rel-L2 <= 2e-3 (measured 4 - 5e-4)
    ck.lt("pc feats rel", rel(f, fr), 2e-3)
    ck.lt("img backbone rel", rel(bbi, bbir), 2e-3)
    ck.lt("img feats rel", rel(fi, fir), 2e-3)
    # per-sample worst case (a single bad cloud must not hide in the batch norm of the error)
    per = ((bb.cpu().double() - bbr.double()).norm(dim=1) / bbr.double().norm(dim=1)).max().item()
    ck.lt("pc backbone worst-sample rel", per, 4e-3)
    ck.done()


TRAIN_FULL = {"c1": 32, "c3": 32, "c4": 16, "ref144": 16}      # c1: half of configs[1]'s 64 pairs (the oracle's four backward passes stay in minutes)
# Gradient-direction floors of the full-batch live-oracle tests (deficit = 1 - cosine against the fp32 oracle, all parameters / worst tensor),
# PER CONFIGURATION (no blanket 0.996).  Linear loss: SURVEY 8c's cosine >= 0.999 everywhere (measured 7.1 - 9.0e-4 over
# rounds 4 - 6, identical to five digits run after run: the masks are seeded; the fp16-emulating oracle itself sits at 6.9 - 8.8e-4, so this IS the
# data format's budget).  Pre-training loss: 1.35 x the larger of (measured, emulated budget) of profiles/r0[4-6]_parity_report.txt --
# c1 1.52 / 1.66e-3 (32 / 64 pairs), c3 1.62e-3, c4 budget 1.23e-3, ref144 budget 3.27e-3 (eight layers of 128 latents at 16 pairs: NOTES round 5).
FLOORS_FULL = {
    "c1": {"linear loss": (1e-3, 1e-2), "NT-Xent loss": (2.2e-3, 2e-2)},
    "c3": {"linear loss": (1e-3, 1e-2), "NT-Xent loss": (2.2e-3, 2e-2)},
    "c4": {"linear loss": (1e-3, 1e-2), "NT-Xent loss": (1.7e-3, 2e-2)},
    "ref144": {"linear loss": (1e-3, 1e-2), "NT-Xent loss": (4.4e-3, 2e-2)},
}


@pytest.mark.parametrize("name", ["c1", "c3", "c4", "ref144"])
def test_full_batch_train_step_loss_and_gradients(name):
    """Train mode at BASELINE batch sizes (c2 architecture at 32 pairs, config 3 at its 32, config 4 at its 16 pairs per GPU),
    dropout 0.1 / 0.5 with the kernels' own masks handed to the oracle.
      * NT-Xent loss: abs <= 5e-3 against the fp32 oracle -- the projection head's BatchNorm sees 64 / 32 samples here;
      * gradients of a loss linear in the backbone features AND of the pre-training loss, every parameter, against the fp32 oracle
        with constant per-configuration floors (FLOORS_FULL: linear loss cosine >= 0.999 everywhere, the pre-training loss >= 0.9978 /
        0.9978 / 0.9983 / 0.9956; measured 0.9991 - 0.9993 / 0.9969 - 0.9994 with fp16 operands, where bf16 operands gave 0.9855 -
        0.9947); the deficit of the fp16-emulating oracle on the same batch is reported beside it."""
    from oracle import torch_oracle as O
    from vipformer_amd import ops
    from tests.test_modules_gpu import budget_check, clear, grads_of
    B = TRAIN_FULL[name]
    ops.rng.seed(4321)
    with ops.rng.pinned():
        pc, im, a = build(name, (0.1, 0.5))
        ck = Checks(f"fullsize-train[{name}, {B} pairs]")
        pts = Hh.synth_points(910, 2 * B, a["N"]); start = Hh.synth_start(910, 2 * B, a["N"])
        imgs = Hh.synth_images(911, B, a["img"], a["img"])
        pc.train(); im.train(); pc.zero_grad(); im.zero_grad()
        with forced_start(start.cuda()):
            f, bb = pc(pts.cuda())
        fi, bbi = im(imgs.cuda())
        f1, f2 = f[:B], f[B:]
        loss = ops.ntxent_loss(f1, f2, 0.1) + ops.ntxent_loss((f1 + f2) / 2, fi, 0.1)
        Rb, Rbi = Hh.synth_like(700, bb.shape), Hh.synth_like(701, bbi.shape)
        ((bb * Rb.cuda()).sum() + (bbi * Rbi.cuda()).sum()).backward(retain_graph=True)
        lin_grads = {id(p): p.grad.clone() for m in (pc, im) for p in m.parameters() if p.grad is not None}
        pc.zero_grad(); im.zero_grad()
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
    fr, bbr = O.pc_forward(s1, pts, start, arch, True, pm, {})
    fir, bbir = O.img_forward(s2, imgs, arch, True, imk, {})
    lref = O.ntxent(fr[:B], fr[B:]) + O.ntxent((fr[:B] + fr[B:]) / 2, fir)
    report(f"fullsize-train[{name}] loss hip {loss.item():.5f} fp32 oracle {lref.item():.5f}")
    ck.lt("pc backbone rel (fp32 oracle)", rel(bb, bbr), 2e-3)          # train mode, dropout 0.1 / 0.5 (measured 3 - 4e-4)
    ck.lt("img backbone rel (fp32 oracle)", rel(bbi, bbir), 2e-3)
    ck.lt("pc feats rel (fp32 oracle)", rel(f, fr), 1e-2)               # behind the BatchNorm head (<= 1e-2; measured 2e-3)
    ck.lt("img feats rel (fp32 oracle)", rel(fi, fir), 1e-2)
    ck.lt("loss abs diff vs fp32 oracle (SURVEY 8c: 5e-3)", abs(loss.item() - lref.item()), 5e-3)
    ((bbr * Rb).sum() + (bbir * Rbi).sum()).backward(retain_graph=True)
    lin_f32 = grads_of((pcp, imp)); clear((pcp, imp))
    lref.backward()
    ntx_f32 = grads_of((pcp, imp)); clear((pcp, imp))
    del fr, bbr, fir, bbir, lref
    with O.emulate_fp16():
        fe, bbe = O.pc_forward(s1, pts, start, arch, True, pm, {})
        fie, bbie = O.img_forward(s2, imgs, arch, True, imk, {})
        le = O.ntxent(fe[:B], fe[B:]) + O.ntxent((fe[:B] + fe[B:]) / 2, fie)
    ck.lt("[emulated] pc backbone rel", rel(bb, bbe), 4e-3)
    ck.lt("[emulated] img backbone rel", rel(bbi, bbie), 4e-3)
    ck.lt("[emulated] loss abs diff", abs(loss.item() - le.item()), 5e-3)
    ((bbe * Rb).sum() + (bbie * Rbi).sum()).backward(retain_graph=True)
    lin_emu = grads_of((pcp, imp)); clear((pcp, imp))
    le.backward()
    ntx_emu = grads_of((pcp, imp)); clear((pcp, imp))
    where = f"fullsize-train[{name}]"
    budget_check(ck, where, "linear loss", pc, im, lambda p: lin_grads.get(id(p)), lin_emu, lin_f32, floors=FLOORS_FULL[name])
    budget_check(ck, where, "NT-Xent loss", pc, im, lambda p: p.grad, ntx_emu, ntx_f32, floors=FLOORS_FULL[name])
    ck.done()


def test_c2_train_step_at_the_benchmarked_batch_64_pairs():
    """BASELINE configs[1] exactly as bench.py runs it: E1CL6SL-H4D256-L96-MR2 at 64 pairs per GPU, train mode,
    dropout 0.1 / 0.5 with the kernels' own masks handed to the oracle -- BatchNorm of the projection head over 128 / 64 rows, 128 x 128
    and 64 x 64 NT-Xent logits.  One fp32 oracle forward + ONE backward (the pre-training loss; the linear-loss and fp16-emulating passes
    run at 32 pairs in test_full_batch_train_step_loss_and_gradients): loss abs <= 5e-3, backbone rel <= 2e-3, features <= 1e-2,
    every parameter's gradient against the configuration's constant floors (FLOORS_FULL["c1"]: all-parameter cosine >= 0.9978, worst tensor >= 0.98)."""
    from oracle import torch_oracle as O
    from vipformer_amd import ops
    from tests.test_modules_gpu import ZERO_GRAD, grads_of
    name, B = "c1", 64
    ops.rng.seed(8765)
    with ops.rng.pinned():
        pc, im, a = build(name, (0.1, 0.5))
        ck = Checks(f"fullsize-train[{name}, {B} pairs]")
        pts = Hh.synth_points(920, 2 * B, a["N"]); start = Hh.synth_start(920, 2 * B, a["N"])
        imgs = Hh.synth_images(921, B, a["img"], a["img"])
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
    fr, bbr = O.pc_forward(s1, pts, start, arch, True, pm, {})
    fir, bbir = O.img_forward(s2, imgs, arch, True, imk, {})
    lref = O.ntxent(fr[:B], fr[B:]) + O.ntxent((fr[:B] + fr[B:]) / 2, fir)
    report(f"fullsize-train[{name}, {B} pairs] loss hip {loss.item():.5f} fp32 oracle {lref.item():.5f}")
    ck.lt("pc backbone rel (fp32 oracle)", rel(bb, bbr), 2e-3)
    ck.lt("img backbone rel (fp32 oracle)", rel(bbi, bbir), 2e-3)
    ck.lt("pc feats rel (fp32 oracle)", rel(f, fr), 1e-2)
    ck.lt("img feats rel (fp32 oracle)", rel(fi, fir), 1e-2)
    ck.lt("loss abs diff vs fp32 oracle (SURVEY 8c: 5e-3)", abs(loss.item() - lref.item()), 5e-3)
    lref.backward()
    g32 = grads_of((pcp, imp))
    rows = []
    for model, gf in ((pc, g32[0]), (im, g32[1])):
        for k, p in model.named_parameters():
            if k.endswith(ZERO_GRAD) or k not in gf or p.grad is None:
                continue
            rows.append((k, p.grad.cpu(), gf[k]))
    cat = lambda i: torch.cat([r[i].reshape(-1) for r in rows])
    d_all = 1 - cosine(cat(1), cat(2))
    per = sorted(((1 - cosine(r[1], r[2]), r[0]) for r in rows), reverse=True)
    for x in per[:3]:
        report(f"fullsize-train[{name}, {B} pairs] [NT-Xent loss] largest deficit vs fp32: hip {x[0]:.5f} {x[1]}")
    report(f"fullsize-train[{name}, {B} pairs] [NT-Xent loss] all-parameter deficit (1 - cos): hip/fp32 {d_all:.5f}")
    f_all, f_min = FLOORS_FULL[name]["NT-Xent loss"]
    ck.lt("[NT-Xent loss] all-parameter gradient deficit (1 - cos) vs fp32", d_all, f_all)
    ck.lt("[NT-Xent loss] worst per-tensor gradient deficit vs fp32", per[0][0], f_min)
    ck.done()


# ---------------------------------------------------------------------------------------------------------------------------------
# Full size against fixtures written by the IMPORTED REFERENCE (tests/golden/make_golden.py make_fullsize): no
# oracle runs on the GPU box's host here.  Dropout 0 on both sides (a fixture cannot replay the kernels' masks; the dropout placement
# is pinned by the dropout_*.npz fixtures and, at full size with the real probabilities, by the live-oracle tests above).
# Floors on the gradient's direction (deficit = 1 - cosine over the strided samples of every parameter, helpers.grad_sample), PER CONFIG and
# DERIVED, not fitted: tests/rounding_budget.py in `fixture` mode runs the fp32 oracle against the same oracle with every fp16 rounding
# point of the HIP data path switched on, on exactly these inputs (profiles/r06_rounding_budget_fixture_fp16_{c1_64,c3_32,c4_16}.txt, row
# "all (the HIP data path)", the strided-sample columns) -- what fp16 operand storage costs on that configuration whatever the kernels do
# (the emulation's FORWARD error equals the kernels' to three digits: features behind the BatchNorm head 5.86e-3 both at c1 / 64 pairs):
#                                   c1 @ 64    c3 @ 32    c4 @ 16       floor                         HIP measured (round 6)
#   linear loss, all parameters     1.9e-4     1.7e-4     1.7e-4        1e-3 = SURVEY 8c's 0.999      1.5e-4 / 1.1e-4 / 0.8e-4
#   NT-Xent, all parameters         6.3e-3     4.3e-3     1.06e-2       2 x the budget                8.2e-3 / 4.2e-3 / 8.9e-3
#   NT-Xent, worst heavy tensor     7.9e-3     5.6e-3     2.1e-2        3 x the budget                1.5e-2 / 7.2e-3 / 4.0e-2
#   NT-Xent, median per tensor      1.2e-3     1.0e-3     0.5e-3        1e-2 (see below)              3.3e-3 / 2.9e-3 / 5.0e-3
# Without dropout the synthetic-weight features of a batch are nearly collinear and the temperature-0.1 softmax is sharp (l_cmid ~ 8.9 against
# ln 63 = 4.1 at chance): dL/dfeats is a small difference of large terms, so the pre-training loss's gradient is far more sensitive to the
# forward roundings here than in the dropout runs of the live-oracle tests above (3e-3 at c1 / 64 pairs), and WHICH tensors take the error
# depends on the realisation of the rounding, not only on its size: per tensor (512-element samples) the kernels sit at 2 - 3 x the emulation
# in the median while the all-parameter figure is within 1.3 x -- the per-tensor floors are therefore guards against a wrong kernel (cosine
# near 0), the all-parameter floor is the parity statement.  The loss scale does not enter (1 / 256 / 4096: same figures to four digits).
# "heavy" = a tensor with >= 64 samples and >= 1 % of the largest gradient norm; the budget's LOWEST cosine over all tensors is meaningless
# (-0.08 .. 0.14: tensors whose gradient is noise).
FIXTURE_FLOORS = {          # name -> {loss tag -> (all-parameter deficit, median per-tensor deficit, worst deficit among heavy tensors)}
    "c1": {"lin": (1e-3, 1e-3, 1e-2), "ntx": (1.3e-2, 1e-2, 2.4e-2)},
    "c3": {"lin": (1e-3, 1e-3, 1e-2), "ntx": (9e-3, 1e-2, 1.7e-2)},
    "c4": {"lin": (1e-3, 1e-3, 1e-2), "ntx": (2.1e-2, 1e-2, 6.3e-2)},
}


@pytest.mark.parametrize("name", ["c1", "c3", "c4"])
def test_full_batch_vs_reference_fixture(name):
    """BASELINE configs[1] / [2] / [3] at their per-GPU batch (64 / 32 / 16 pairs) against fullsize_<name>.npz, which the reference
    itself wrote (partseg.py:527-550,661-680 forward, autograd backward; pretrain.py:189-207 loss):
      * eval mode: backbone + projected features of both models, rel-L2 <= 2e-3 (SURVEY 8c, fp16 operands);
      * train mode (dropout 0): backbone <= 2e-3, features behind the BatchNorm head <= 1e-2, pre-training loss abs <= 5e-3,
        BatchNorm running statistics <= 3e-2;
      * gradients of the linear loss and of the pre-training loss, every parameter: norm within 8 % of the reference's for every tensor
        that carries >= 0.1 % of the largest norm, direction by the strided samples against FIXTURE_FLOORS (derived per configuration from
        the fp16 rounding budget of exactly these inputs)."""
    import json
    import os
    from vipformer_amd import ops
    from tests.test_modules_gpu import ZERO_GRAD
    g = Hh.golden(f"fullsize_{name}.npz")
    B = int(g["meta"][0])
    assert B == Hh.FULL_BATCH[name]
    pc, im, a = build(name)
    ck = Checks(f"fullsize-fixture[{name}, {B} pairs]")
    pts = Hh.synth_points(int(g["meta"][1]), 2 * B, a["N"]).cuda(); start = Hh.synth_start(int(g["meta"][1]), 2 * B, a["N"]).cuda()
    imgs = Hh.synth_images(int(g["meta"][2]), B, a["img"], a["img"]).cuda()
    pc.eval(); im.eval()
    with torch.no_grad(), forced_start(start):
        f, bb = pc(pts)
        fi, bbi = im(imgs)
    ck.lt("pc eval backbone rel", rel(bb, g["pc_eval_backbone"]), 2e-3)
    ck.lt("pc eval feats rel", rel(f, g["pc_eval_feats"]), 2e-3)
    ck.lt("img eval backbone rel", rel(bbi, g["img_eval_backbone"]), 2e-3)
    ck.lt("img eval feats rel", rel(fi, g["img_eval_feats"]), 2e-3)
    ref_bb = torch.from_numpy(g["pc_eval_backbone"]).double()
    per = ((bb.cpu().double() - ref_bb).norm(dim=1) / ref_bb.norm(dim=1)).max().item()
    ck.lt("pc eval backbone worst-sample rel", per, 4e-3)
    pc.train(); im.train(); pc.zero_grad(); im.zero_grad()
    with forced_start(start):
        f, bb = pc(pts)
    fi, bbi = im(imgs)
    ck.lt("pc train backbone rel", rel(bb, g["pc_train_backbone"]), 2e-3)
    ck.lt("img train backbone rel", rel(bbi, g["img_train_backbone"]), 2e-3)
    ck.lt("pc train feats rel", rel(f, g["pc_train_feats"]), 1e-2)
    ck.lt("img train feats rel", rel(fi, g["img_train_feats"]), 1e-2)
    f1, f2 = f[:B], f[B:]
    l_im = ops.ntxent_loss(f1, f2, 0.1); l_cm = ops.ntxent_loss((f1 + f2) / 2, fi, 0.1)
    total = l_im + 1.0 * l_cm
    got = np.array([total.item(), l_im.item(), l_cm.item()])
    report(f"fullsize-fixture[{name}] loss hip {got} reference {g['loss']}")
    ck.lt("loss abs diff vs the reference (SURVEY 8c: 5e-3)", float(np.abs(got - g["loss"]).max()), 5e-3)
    for k in ("latent_head.0.running_mean", "latent_head.0.running_var", "group2emb.first_conv.1.running_var"):
        ck.lt(f"buffer {k} rel", rel(pc.state_dict()[k], g["pc_buf." + k]), 3e-2)
    names = json.load(open(os.path.join(Hh.GOLDEN_DIR, f"grad_names_{name}.json")))
    lin = (bb * Hh.synth_like(700, bb.shape).cuda()).sum() + (bbi * Hh.synth_like(701, bbi.shape).cuda()).sum()
    # scaler.scale(loss).backward() (pretrain.py:209): the backward kernels take fp16 gradient operands, so -- as in the reference's autocast
    # loop -- the loss is scaled up front and the gradients are read back divided by the scale (tests/conftest.py: the backed-off scale 256)
    SCALE = float(os.environ.get("VPF_FIXTURE_LOSS_SCALE", Hh.TEST_LOSS_SCALE))
    for tag, loss in (("lin", lin), ("ntx", total)):
        pc.zero_grad(); im.zero_grad()
        (loss * SCALE).backward(retain_graph=(tag == "lin"))
        got_s, ref_s, worst, per = [], [], (0.0, "-"), []
        for which, model in (("pc", pc), ("img", im)):
            params = dict(model.named_parameters())
            refn, refs = g[f"{which}_{tag}_norms"], torch.from_numpy(g[f"{which}_{tag}_samples"])
            off = 0
            dev = 0.0
            for i, k in enumerate(names[which]):
                p = params[k]
                gk = p.grad / SCALE if p.grad is not None else torch.zeros_like(p)
                smp = Hh.grad_sample(gk)
                r = refs[off:off + smp.numel()]; off += smp.numel()
                if k.endswith(ZERO_GRAD) or refn[i] <= 1e-3 * refn.max():
                    continue                                     # exactly zero by BatchNorm's shift invariance / carries no weight
                dev = max(dev, abs(gk.double().norm().item() / refn[i] - 1.0))
                got_s.append(smp); ref_s.append(r)
                if smp.numel() >= 64:
                    d = 1 - cosine(smp, r)
                    per.append(d)
                    if refn[i] >= 1e-2 * refn.max():
                        worst = max(worst, (d, f"{which}.{k}"))
            assert off == refs.numel()
            ck.lt(f"[{tag}] {which} gradient-norm ratio, max deviation from 1", dev, 0.08)
        d_all = 1 - cosine(torch.cat(got_s), torch.cat(ref_s))
        med = float(np.median(per))
        report(f"fullsize-fixture[{name}] [{tag}] sampled all-parameter deficit {d_all:.5f}; per tensor: median {med:.5f}, worst heavy tensor {worst[0]:.5f} {worst[1]}")
        f_all, f_med, f_min = FIXTURE_FLOORS[name][tag]
        ck.lt(f"[{tag}] all-parameter gradient deficit (1 - cos, strided samples) vs the reference", d_all, f_all)
        ck.lt(f"[{tag}] median per-tensor gradient deficit vs the reference", med, f_med)
        ck.lt(f"[{tag}] worst gradient deficit among tensors with >= 1 % of the largest norm vs the reference", worst[0], f_min)
    ck.done()
