"""GPU parity of the mirrored modules (vipformer_amd.model.pointcloud.*) against
  (a) the golden fixtures captured from the imported reference (fp32 torch-CPU), and
  (b) the oracle run live on the host with the SAME dropout masks the kernels drew.

Precision contract under test: bf16 MFMA operands / fp32 accumulate / fp32 residual stream
vs the fp32 reference.  Stated tolerances: forward activations rel-L2 <= 1e-2,
loss abs <= 5e-3 (relative to |loss| ~ 3), gradients cosine >= 0.999 (per tensor, for
tensors whose reference norm is not negligible) and rel-L2 <= 3e-2.
"""
import json
import os

import numpy as np
import pytest
import torch

from tests import helpers as Hh

pytestmark = pytest.mark.gpu
FWD_TOL = 1e-2
GRAD_TOL = 3e-2
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.txt")


def report(line):
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def rel(a, b):
    a = torch.as_tensor(np.asarray(a.detach().float().cpu()) if torch.is_tensor(a) else a).double().flatten()
    b = torch.as_tensor(np.asarray(b.detach().float().cpu()) if torch.is_tensor(b) else b).double().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def cosine(a, b):
    a = torch.as_tensor(np.asarray(a.detach().float().cpu()) if torch.is_tensor(a) else a).double().flatten()
    b = torch.as_tensor(np.asarray(b.detach().float().cpu()) if torch.is_tensor(b) else b).double().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


class forced_start:
    """Make farthest_point_sample's torch.randint (utils.py:71) return ``start``."""

    def __init__(self, start):
        self.start = start

    def __enter__(self):
        self.real = torch.randint
        torch.randint = lambda *a, **k: self.start.clone()

    def __exit__(self, *a):
        torch.randint = self.real


def build(name, drops=(0.0, 0.0)):
    from vipformer_amd.model.pointcloud import CrossFormer_img_mp, CrossFormer_pc_mp, PointCloudInputAdapter
    a = Hh.ARCHS[name]
    ad = PointCloudInputAdapter((a["N"], 3), a["D"])
    pc = CrossFormer_pc_mp(ad, a["G"], a["D"], a["K"], 1, a["H"], a["S"], a["H"], a["MR"], 0.0, drops[0], drops[1], True)
    im = CrossFormer_img_mp(a["img"], a["img"], a["patch"], a["D"], 1, a["H"], a["S"], a["H"], a["MR"], 0.0, drops[0], drops[1], True)
    pc.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pc_{name}.json"), 100))
    im.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes(f"keys_img_{name}.json"), 200))
    return pc.cuda(), im.cuda(), a


@pytest.mark.parametrize("name", ["tiny", "tiny2", "c1"])
def test_stages_vs_reference_golden(name):
    from vipformer_amd.model.pointcloud import utils as U
    pc, im, a = build(name)
    g = Hh.golden(f"stages_{name}.npz")
    B = 2
    c1 = name == "c1"
    pts = Hh.synth_points(300, 2 * B, a["N"]).cuda()
    start = Hh.synth_start(300, 2 * B, a["N"]).cuda()
    with forced_start(start):
        nb, ct = U.divide_patches(pts, a["G"], a["K"])
    pc.train()
    y = pc.group2emb(nb)
    R = Hh.synth_like(500, y.shape).cuda()
    pc.zero_grad()
    (y * R).sum().backward()
    r = rel(y[:, :8] if c1 else y, g["g2e_train"])
    report(f"{name} group2emb train fwd rel {r:.2e}")
    assert r < FWD_TOL
    g2e = pc.group2emb
    assert np.allclose(g2e.first_conv[1].running_mean.cpu(), g["g2e_rm1"], rtol=1e-3, atol=1e-4)
    assert np.allclose(g2e.first_conv[1].running_var.cpu(), g["g2e_rv1"], rtol=1e-3, atol=1e-4)
    assert np.allclose(g2e.second_conv[1].running_mean.cpu(), g["g2e_rm2"], rtol=2e-2, atol=2e-3)
    assert np.allclose(g2e.second_conv[1].running_var.cpu(), g["g2e_rv2"], rtol=2e-2, atol=2e-3)
    for k, p in g2e.named_parameters():
        ref = g["g2e_grad." + k]
        got = p.grad.reshape(-1)[:ref.size].reshape(ref.shape) if c1 and p.numel() > 4096 else p.grad
        rr, cc = rel(got, ref), cosine(got, ref)
        report(f"{name} group2emb grad {k}: rel {rr:.2e} cos {cc:.5f} |ref| {np.linalg.norm(ref):.2e}")
        if np.linalg.norm(ref) > 1e-3 * max(1.0, ref.size ** 0.5 * 1e-2):   # conv biases ahead of a BatchNorm have ~0 gradient
            assert cc > 0.999 and rr < GRAD_TOL, (k, rr, cc)
    pc.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pc_{name}.json"), 100))
    pc.eval()
    with torch.no_grad():
        ye = pc.group2emb(nb)
        assert rel(ye[:, :8] if c1 else ye, g["g2e_eval"]) < FWD_TOL
        ra = rel(pc.input_adapter(pts)[:, :32], g["adapter"])
        from vipformer_amd import ops
        ps = ops.PosMLPFn.apply(ct, pc.position_emb, *pc.position_emb.parameters())
        rp = rel(ps[:, :8] if c1 else ps, g["pos"])
        report(f"{name} adapter rel {ra:.2e} pos rel {rp:.2e}")
        assert ra < FWD_TOL and rp < FWD_TOL
    # cross-attention layer and one self-attention layer, forward + input grads + weight grads
    pc.train()
    enc = pc.encoder
    x = Hh.synth_like(600, (2 * B, a["G"], a["D"])).cuda().requires_grad_()
    kv = Hh.synth_like(601, (2 * B, a["N"], a["D"])).cuda().requires_grad_()
    pc.zero_grad()
    yca = enc.cross_attn_1(x, kv, None)
    Rl = Hh.synth_like(602, yca.shape).cuda()
    (yca * Rl).sum().backward()
    sl = (slice(None), slice(0, 8)) if c1 else (slice(None),)
    for nm, got, ref in (("ca_out", yca[sl], g["ca_out"]), ("ca_dx", x.grad[sl], g["ca_dx"]), ("ca_dkv", kv.grad[:, :32], g["ca_dkv"])):
        rr = rel(got, ref)
        report(f"{name} {nm} rel {rr:.2e}")
        assert rr < (FWD_TOL if nm == "ca_out" else GRAD_TOL), (nm, rr)
    for k, p in enc.cross_attn_n.named_parameters():
        ref = g["ca_grad." + k]
        got = p.grad.reshape(-1)[:ref.size].reshape(ref.shape) if c1 and p.numel() > 4096 else p.grad
        rr, cc = rel(got, ref), cosine(got, ref)
        report(f"{name} ca grad {k}: rel {rr:.2e} cos {cc:.5f}")
        assert cc > 0.999 and rr < GRAD_TOL, (k, rr, cc)
    x2 = Hh.synth_like(603, (2 * B, a["G"], a["D"])).cuda().requires_grad_()
    pc.zero_grad()
    ysa = enc.sa_layers[0](x2)
    (ysa * Rl).sum().backward()
    assert rel(ysa[sl], g["sa_out"]) < FWD_TOL and rel(x2.grad[sl], g["sa_dx"]) < GRAD_TOL
    for k, p in enc.sa_layers[0].named_parameters():
        ref = g["sa_grad." + k]
        got = p.grad.reshape(-1)[:ref.size].reshape(ref.shape) if c1 and p.numel() > 4096 else p.grad
        rr, cc = rel(got, ref), cosine(got, ref)
        report(f"{name} sa grad {k}: rel {rr:.2e} cos {cc:.5f}")
        assert cc > 0.999 and rr < GRAD_TOL, (k, rr, cc)


@pytest.mark.parametrize("name", ["tiny", "tiny2", "c1"])
def test_models_vs_reference_golden(name):
    from vipformer_amd import ops
    pc, im, a = build(name)
    g = Hh.golden(f"model_{name}.npz")
    B = 2
    pts = Hh.synth_points(300, 2 * B, a["N"]).cuda()
    start = Hh.synth_start(300, 2 * B, a["N"]).cuda()
    imgs = Hh.synth_images(400, B, a["img"], a["img"]).cuda()
    pc.eval(); im.eval()
    with torch.no_grad(), forced_start(start):
        f, bb = pc(pts)
        fi, bbi = im(imgs)
    for nm, got, ref in (("pc_eval_feats", f, g["pc_eval_feats"]), ("pc_eval_backbone", bb, g["pc_eval_backbone"]),
                         ("img_eval_feats", fi, g["img_eval_feats"]), ("img_eval_backbone", bbi, g["img_eval_backbone"])):
        rr = rel(got, ref)
        report(f"{name} {nm} rel {rr:.2e}")
        assert rr < 2 * FWD_TOL, (nm, rr)           # whole network: 7-9 layers of bf16 rounding
    pc.train(); im.train(); pc.zero_grad(); im.zero_grad()
    with forced_start(start):
        f, bb = pc(pts)
    fi, bbi = im(imgs)
    f1, f2 = f[:B], f[B:]
    l_im = ops.ntxent_loss(f1, f2, 0.1)
    l_cm = ops.ntxent_loss((f1 + f2) / 2, fi, 0.1)
    loss = l_im + l_cm
    loss.backward()
    got = np.array([loss.item(), l_im.item(), l_cm.item()])
    report(f"{name} loss got {got} ref {g['loss']}")
    assert np.allclose(got, g["loss"], atol=2e-2), (got, g["loss"])
    names = json.load(open(os.path.join(Hh.GOLDEN_DIR, f"grad_names_{name}.json")))
    for which, model, key in (("pc", pc, "pc_grad"), ("img", im, "img_grad")):
        params = dict(model.named_parameters())
        norms = np.array([params[k].grad.double().norm().item() if params[k].grad is not None else 0.0 for k in names[which]])
        refn = g[key + "_norms"]
        big = refn > 1e-3 * refn.max()
        ratio = norms[big] / refn[big]
        report(f"{name} {which} grad-norm ratio min {ratio.min():.3f} max {ratio.max():.3f}")
        assert np.all(np.abs(ratio - 1) < 0.08), list(zip(np.array(names[which])[big], ratio))
        heads = np.stack([torch.cat([params[k].grad.reshape(-1)[:8].cpu(), torch.zeros(max(0, 8 - params[k].numel()))]).numpy()
                          for k in names[which]])
        cc = cosine(torch.from_numpy(heads[big]), g[key + "_heads"][big])
        report(f"{name} {which} grad-heads cosine {cc:.5f}")
        assert cc > 0.995
    for k in ("latent_head.0.running_mean", "latent_head.0.running_var", "group2emb.first_conv.1.running_var"):
        assert np.allclose(pc.state_dict()[k].cpu().numpy(), g["pc_buf." + k], rtol=3e-2, atol=3e-3), k


def _site_masks(model, B_tokens, kv_len, a, device):
    """Export, from the kernels' RNG, the keep mask of every dropout site of ``model.encoder`` under the
    oracle's site names."""
    from vipformer_amd import ops
    H, D = a["H"], a["D"]
    table = {}
    enc = model.encoder
    Bq, Lq = B_tokens

    def layer(layer_mod, tag, Lkv, p_att, p1, p2):
        att = layer_mod[0].module.attention
        table[tag + ".attn"] = ops.dropout_keep_mask(att.site_attn, p_att, (Bq * H, Lq, Lkv), device).float().cpu()
        table[tag + ".res1"] = ops.dropout_keep_mask(layer_mod[0].site, p1, (Bq, Lq, D), device).float().cpu()
        table[tag + ".res2"] = ops.dropout_keep_mask(layer_mod[1].site, p2, (Bq, Lq, D), device).float().cpu()

    p_att, p_mlp = enc.cross_attn_1[0].module.attention.dropout.p, enc.cross_attn_1[1].dropout.p
    layer(enc.cross_attn_1, "ca", kv_len, p_att, enc.cross_attn_1[0].dropout.p, p_mlp)
    for i, sa in enumerate(enc.sa_layers):
        layer(sa, f"sa{i}", Lq, p_att, sa[0].dropout.p, sa[1].dropout.p)
    return table


@pytest.mark.parametrize("name", ["tiny", "tiny2", "c1"])
def test_training_step_with_dropout_vs_oracle(name):
    """Train mode with the real dropout probabilities (0.1 / 0.5): the kernels' own masks are exported and
    handed to the oracle, so forward, loss and gradients must agree within the bf16 tolerances."""
    from oracle import torch_oracle as O
    from vipformer_amd import ops
    ops.rng.seed(1234)
    pc, im, a = build(name, (0.1, 0.5))
    B = 2
    pts = Hh.synth_points(300, 2 * B, a["N"]); start = Hh.synth_start(300, 2 * B, a["N"])
    imgs = Hh.synth_images(400, B, a["img"], a["img"])
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
    pc_sd = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pc_{name}.json"), 100)
    im_sd = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_img_{name}.json"), 200)
    pcp = {k: v.clone().requires_grad_() for k, v in pc_sd.items() if v.dtype == torch.float32 and "running" not in k and "cross_attn_1." not in k}
    imp = {k: v.clone().requires_grad_() for k, v in im_sd.items() if v.dtype == torch.float32 and "running" not in k and "cross_attn_1." not in k}
    s1 = dict(pc_sd); s1.update(pcp); s2 = dict(im_sd); s2.update(imp)
    for s in (s1, s2):
        for k in list(s):
            if "cross_attn_1." in k:
                s[k] = s[k.replace("cross_attn_1.", "cross_attn_n.")]
    lref, _, _ = O.pretrain_losses(s1, s2, pts[:B], pts[B:], imgs, start, arch, True, pm, imk, {}, {})
    lref.backward()
    report(f"{name} dropout step: loss hip {loss.item():.5f} oracle {lref.item():.5f}")
    assert abs(loss.item() - lref.item()) < 3e-2
    worst = (1.0, "")
    for model, ref in ((pc, pcp), (im, imp)):
        for k, p in model.named_parameters():
            r = ref[k].grad
            if r is None or r.norm() < 1e-3 * max(1.0, r.numel() ** 0.5 * 1e-2):
                continue
            cc = cosine(p.grad, r)
            if cc < worst[0]:
                worst = (cc, k)
    report(f"{name} dropout step: worst grad cosine {worst[0]:.5f} at {worst[1]}")
    assert worst[0] > 0.99, worst
