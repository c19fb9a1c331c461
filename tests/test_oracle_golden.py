"""CPU: pins the oracle (oracle/vpf_oracle.c + oracle/torch_oracle.py) against the
fixtures captured from the imported reference (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from tests import helpers as Hh
from oracle import torch_oracle as O

PRE = ["u1024", "u2048", "d1024", "g1024", "c6_1000", "u256", "u512"]


@pytest.mark.parametrize("tag", PRE)
def test_preproc_bit_exact(tag):
    g = Hh.golden(f"preproc_{tag}.npz")
    seed, B, N, C, G, K = [int(v) for v in g["meta"]]
    mode = {"d": "dups", "g": "grid"}.get(tag[0], "uniform")
    pts = Hh.synth_points(seed, B, N, C, mode)
    start = Hh.synth_start(seed, B, N)
    idx = O.fps_indices(pts, start, G)
    assert np.array_equal(idx.numpy(), g["fps_idx"])                      # FPS: bit-exact indices
    nb, ct, kidx = O.divide_patches(pts, idx, K, True)
    assert np.array_equal(ct.numpy(), g["centers"])
    d = O.square_distance(ct, pts).numpy().view(np.uint32)
    gb = g["sqdist_bits"]
    assert np.array_equal(d[:, :gb.shape[1], :], gb)                      # distances: bitwise
    assert np.array_equal(kidx.numpy(), g["knn_canonical"])               # canonical order
    if bool(g["tie_free"]):                                               # vs raw topk: same SET
        assert np.array_equal(np.sort(kidx.numpy(), -1), g["knn_raw_sorted"])
    assert np.array_equal(nb.numpy(), g["neighbors"])                     # incl. the :36 axis quirk
    # quirk sanity: members 0..2 centred, the rest absolute
    raw = torch.gather(pts.unsqueeze(1).expand(B, G, N, C), 2, kidx.unsqueeze(-1).expand(B, G, K, C))
    assert torch.equal(nb[:, :, 3:], raw[:, :, 3:]) and torch.equal(nb[:, :, :3], raw[:, :, :3] - ct.unsqueeze(2))


def _arch(name, drops=(0.0, 0.0)):
    a = Hh.ARCHS[name]
    return O.Arch(D=a["D"], H=a["H"], G=a["G"], K=a["K"], S=a["S"], MR=a["MR"], img=a["img"], patch=a["patch"],
                  atten_drop=drops[0], mlp_drop=drops[1]), a


def _close(x, ref, rtol=2e-5, atol=2e-5):
    x = x.detach().numpy() if torch.is_tensor(x) else x
    np.testing.assert_allclose(x, ref, rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", ["tiny", "tiny2", "c1", "c3", "c4", "ref144", "ref144m4"])
def test_models_vs_reference(name):
    arch, a = _arch(name)
    g = Hh.golden(f"model_{name}.npz")
    B = Hh.MODEL_BATCH[name]
    pc = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pc_{name}.json"), 100)
    im = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_img_{name}.json"), 200)
    pts = Hh.synth_points(300, 2 * B, a["N"]); start = Hh.synth_start(300, 2 * B, a["N"])
    imgs = Hh.synth_images(400, B, a["img"], a["img"])
    with torch.no_grad():
        f, bb = O.pc_forward(pc, pts, start, arch, False)
        fi, bbi = O.img_forward(im, imgs, arch, False)
    _close(f, g["pc_eval_feats"]); _close(bb, g["pc_eval_backbone"])
    _close(fi, g["img_eval_feats"]); _close(bbi, g["img_eval_backbone"])
    # train mode (dropout 0) + loss + grads
    names = json.load(open(os.path.join(Hh.GOLDEN_DIR, f"grad_names_{name}.json")))
    pcp = {k: pc[k].clone().requires_grad_() for k in names["pc"]}
    imp = {k: im[k].clone().requires_grad_() for k in names["img"]}
    pc_sd = dict(pc); pc_sd.update(pcp)
    im_sd = dict(im); im_sd.update(imp)
    for k in list(pc_sd):
        if "cross_attn_1." in k:
            pc_sd[k] = pc_sd[k.replace("cross_attn_1.", "cross_attn_n.")]
    for k in list(im_sd):
        if "cross_attn_1." in k:
            im_sd[k] = im_sd[k.replace("cross_attn_1.", "cross_attn_n.")]
    bufs = {}
    f, bb = O.pc_forward(pc_sd, pts, start, arch, True, O.Masks("off"), bufs)
    fi, bbi = O.img_forward(im_sd, imgs, arch, True, O.Masks("off"), {})
    l_im = O.ntxent(f[:B], f[B:]); l_cm = O.ntxent((f[:B] + f[B:]) / 2, fi)
    _close(np.array([(l_im + l_cm).item(), l_im.item(), l_cm.item()]), g["loss"], 1e-5, 1e-5)
    _close(f, g["pc_train_feats"], 1e-4, 1e-4); _close(fi, g["img_train_feats"], 1e-4, 1e-4)
    (bb * Hh.synth_like(700, bb.shape)).sum().backward()
    (bbi * Hh.synth_like(701, bbi.shape)).sum().backward()
    zero = lambda p: p.grad if p.grad is not None else torch.zeros_like(p)
    for which, params, key in (("pc", pcp, "pc_grad"), ("img", imp, "img_grad")):
        norms = np.array([zero(params[k]).double().norm().item() for k in names[which]])
        np.testing.assert_allclose(norms, g[key + "_norms"], rtol=2e-3, atol=1e-5)
        heads = np.stack([torch.cat([zero(params[k]).reshape(-1)[:8], torch.zeros(max(0, 8 - params[k].numel()))]).numpy()
                          for k in names[which]])
        np.testing.assert_allclose(heads, g[key + "_heads"], rtol=5e-3, atol=2e-5)
    for k in ("latent_head.0.running_mean", "latent_head.0.running_var", "group2emb.first_conv.1.running_var"):
        _close(bufs[k], g["pc_buf." + k])


@pytest.mark.parametrize("name", ["c1", "c3", "c4"])
def test_oracle_at_full_batch_vs_reference(name):
    """The restatement at BASELINE's per-GPU batches (configs[1]: 64 pairs, [2]: 32, [3]: 16) against fullsize_<name>.npz, written by the imported reference
    (make_golden.py make_fullsize): eval + train-mode (dropout 0) features, the pre-training loss and BOTH losses' gradients (norms and
    the strided samples) -- what the live-oracle GPU tests at full size (tests/test_fullsize_gpu.py, dropout masks replayed) lean on.
    The HIP path is compared with the same three fixtures directly on the GPU box (test_full_batch_vs_reference_fixture)."""
    arch, a = _arch(name)
    g = Hh.golden(f"fullsize_{name}.npz")
    B = int(g["meta"][0])
    pc = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pc_{name}.json"), 100)
    im = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_img_{name}.json"), 200)
    pts = Hh.synth_points(int(g["meta"][1]), 2 * B, a["N"]); start = Hh.synth_start(int(g["meta"][1]), 2 * B, a["N"])
    imgs = Hh.synth_images(int(g["meta"][2]), B, a["img"], a["img"])
    with torch.no_grad():
        f, bb = O.pc_forward(pc, pts, start, arch, False)
        fi, bbi = O.img_forward(im, imgs, arch, False)
    _close(f, g["pc_eval_feats"]); _close(bb, g["pc_eval_backbone"])
    _close(fi, g["img_eval_feats"]); _close(bbi, g["img_eval_backbone"])
    names = json.load(open(os.path.join(Hh.GOLDEN_DIR, f"grad_names_{name}.json")))
    pcp = {k: pc[k].clone().requires_grad_() for k in names["pc"]}
    imp = {k: im[k].clone().requires_grad_() for k in names["img"]}
    pc_sd = dict(pc); pc_sd.update(pcp)
    im_sd = dict(im); im_sd.update(imp)
    for sd in (pc_sd, im_sd):
        for k in list(sd):
            if "cross_attn_1." in k:
                sd[k] = sd[k.replace("cross_attn_1.", "cross_attn_n.")]
    f, bb = O.pc_forward(pc_sd, pts, start, arch, True, O.Masks("off"), {})
    fi, bbi = O.img_forward(im_sd, imgs, arch, True, O.Masks("off"), {})
    l_im = O.ntxent(f[:B], f[B:]); l_cm = O.ntxent((f[:B] + f[B:]) / 2, fi)
    total = l_im + l_cm
    _close(np.array([total.item(), l_im.item(), l_cm.item()]), g["loss"], 1e-5, 1e-5)
    _close(f, g["pc_train_feats"], 1e-4, 1e-4); _close(fi, g["img_train_feats"], 1e-4, 1e-4)
    _close(bb, g["pc_train_backbone"], 1e-4, 1e-4); _close(bbi, g["img_train_backbone"], 1e-4, 1e-4)
    lin = (bb * Hh.synth_like(700, bb.shape)).sum() + (bbi * Hh.synth_like(701, bbi.shape)).sum()
    for tag, loss in (("lin", lin), ("ntx", total)):
        for d in (pcp, imp):
            for v in d.values():
                v.grad = None
        loss.backward(retain_graph=(tag == "lin"))
        for which, params in (("pc", pcp), ("img", imp)):
            zero = lambda p: p.grad if p.grad is not None else torch.zeros_like(p)
            norms = np.array([zero(params[k]).double().norm().item() for k in names[which]])
            np.testing.assert_allclose(norms, g[f"{which}_{tag}_norms"], rtol=2e-3, atol=1e-5 * g[f"{which}_{tag}_norms"].max())
            smp = torch.cat([Hh.grad_sample(zero(params[k])) for k in names[which]])
            ref = torch.from_numpy(g[f"{which}_{tag}_samples"])
            cos = float((smp.double() @ ref.double()) / (smp.double().norm() * ref.double().norm()))
            assert cos > 1 - 1e-6, (tag, which, cos)
            np.testing.assert_allclose(smp.numpy(), ref.numpy(), rtol=2e-2, atol=1e-4 * float(ref.abs().max()))


@pytest.mark.parametrize("name", ["tiny", "tiny2", "c1", "c3", "c4", "ref144", "ref144m4"])
def test_stages_vs_reference(name):
    arch, a = _arch(name)
    g = Hh.golden(f"stages_{name}.npz")
    B = 2
    sd = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pc_{name}.json"), 100)
    pts = Hh.synth_points(300, 2 * B, a["N"]); start = Hh.synth_start(300, 2 * B, a["N"])
    nb, ct, _ = O.divide_patches(pts, O.fps_indices(pts, start, arch.G), arch.K)
    pre = "group2emb."
    wk = [k for k in sd if k.startswith(pre) and not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    p = {k: sd[k].clone().requires_grad_() for k in wk}
    s2 = dict(sd); s2.update(p)
    bufs = {}
    y = O.group2emb(s2, pre, nb, True, bufs)
    R = Hh.synth_like(500, y.shape)
    (y * R).sum().backward()
    c1 = name in Hh.FULLSIZE          # full-size fixtures hold slices
    _close(y[:, :8] if c1 else y, g["g2e_train"], 1e-4, 1e-4)
    _close(bufs[pre + "first_conv.1.running_mean"], g["g2e_rm1"]); _close(bufs[pre + "first_conv.1.running_var"], g["g2e_rv1"])
    _close(bufs[pre + "second_conv.1.running_mean"], g["g2e_rm2"]); _close(bufs[pre + "second_conv.1.running_var"], g["g2e_rv2"])
    for k in wk:
        ref = g["g2e_grad." + k[len(pre):]]
        got = p[k].grad.reshape(-1)[:ref.size].reshape(ref.shape) if c1 and p[k].numel() > 4096 else p[k].grad
        np.testing.assert_allclose(got.numpy(), ref, rtol=2e-3, atol=2e-3 * float(np.abs(ref).max()))
    with torch.no_grad():
        ye = O.group2emb(sd, pre, nb, False)
        _close(ye[:, :8] if c1 else ye, g["g2e_eval"], 1e-4, 1e-4)
        _close(O.adapter(sd, "input_adapter.", pts)[:, :32], g["adapter"], 1e-4, 1e-5)
        ps = O.pos_mlp(sd, "position_emb.", ct)
        _close(ps[:, :8] if c1 else ps, g["pos"], 1e-4, 1e-5)
    # CA / SA layers fwd + input grads
    x = Hh.synth_like(600, (2 * B, arch.G, arch.D)).requires_grad_()
    kv = Hh.synth_like(601, (2 * B, a["N"], arch.D)).requires_grad_()
    m = O.Masks("off")
    yca = O.ca_layer(sd, "encoder.cross_attn_1.", x, kv, arch, m, "ca")
    Rl = Hh.synth_like(602, yca.shape)
    (yca * Rl).sum().backward()
    sl = (slice(None), slice(0, 8)) if c1 else (slice(None),)
    _close(yca[sl], g["ca_out"], 1e-4, 1e-4); _close(x.grad[sl], g["ca_dx"], 1e-3, 1e-4)
    _close(kv.grad[:, :32], g["ca_dkv"], 1e-3, 1e-4)
    x2 = Hh.synth_like(603, (2 * B, arch.G, arch.D)).requires_grad_()
    ysa = O.sa_layer(sd, "encoder.sa_layers.0.", x2, arch, m, "sa0")
    (ysa * Rl).sum().backward()
    _close(ysa[sl], g["sa_out"], 1e-4, 1e-4); _close(x2.grad[sl], g["sa_dx"], 1e-3, 1e-4)


@pytest.mark.parametrize("name", ["tiny", "tiny2", "ref144", "ref144m4"])
def test_dropout_placement_vs_reference(name):
    """Train mode with the real probabilities: the oracle draws its masks from torch's
    RNG in the reference's call order, so equal seeds must give equal outputs.  Pins
    WHERE each dropout sits and with which p (partseg.py:165-166,186-187)."""
    arch, a = _arch(name, (0.1, 0.5))
    g = Hh.golden(f"dropout_{name}.npz")
    B = Hh.MODEL_BATCH[name]
    pc = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pc_{name}.json"), 100)
    im = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_img_{name}.json"), 200)
    pts = Hh.synth_points(300, 2 * B, a["N"]); start = Hh.synth_start(300, 2 * B, a["N"])
    imgs = Hh.synth_images(400, B, a["img"], a["img"])
    with torch.no_grad():
        torch.manual_seed(77)
        f, bb = O.pc_forward(pc, pts, start, arch, True, O.Masks("torch"), {})
        torch.manual_seed(78)
        fi, bbi = O.img_forward(im, imgs, arch, True, O.Masks("torch"), {})
    _close(f, g["pc_feats"], 1e-4, 1e-4); _close(bb, g["pc_backbone"], 1e-4, 1e-4)
    _close(fi, g["img_feats"], 1e-4, 1e-4); _close(bbi, g["img_backbone"], 1e-4, 1e-4)


def test_param_counts_match_paper_tables():
    """assets/tab1.png / tab2.png: 5.1 M and 16.7 M parameters."""
    c = json.load(open(os.path.join(Hh.GOLDEN_DIR, "param_counts.json")))
    assert c["c1"]["pc_params"] == 4074368 and c["c3"]["pc_params"] == 5127040 and c["c4"]["pc_params"] == 16654336
    assert c["c1"]["pc_state"] == 148 and c["c1"]["img_state"] == 123 and c["c3"]["pc_state"] == 174


@pytest.mark.parametrize("name", ["tiny", "c1"])
def test_finetune_classifier_vs_reference(name):
    """oracle.pc_ft_forward vs CrossFormer_pc_mp_ft of the reference (partseg.py:553-605; fixture: make_golden.py make_ft)."""
    arch, a = _arch(name)
    g = Hh.golden(f"modelft_{name}.npz")
    B = Hh.MODEL_BATCH[name]
    sd = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pcft_{name}.json"), 100)
    pts = Hh.synth_points(300, 2 * B, a["N"]); start = Hh.synth_start(300, 2 * B, a["N"])
    with torch.no_grad():
        _close(O.pc_ft_forward(sd, pts, start, arch, False), g["eval_logits"], 1e-4, 1e-4)
    names = json.load(open(os.path.join(Hh.GOLDEN_DIR, f"grad_names_pcft_{name}.json")))
    hp = {k: sd[k].clone().requires_grad_() for k in names}
    sd2 = dict(sd); sd2.update(hp)
    for k in list(sd2):
        if "cross_attn_1." in k:
            sd2[k] = sd2[k.replace("cross_attn_1.", "cross_attn_n.")]
    bufs = {}
    y = O.pc_ft_forward(sd2, pts, start, arch, True, O.Masks("off"), bufs)
    _close(y, g["train_logits"], 2e-4, 2e-4)
    (y * Hh.synth_like(710, y.shape)).sum().backward()
    norms = np.array([hp[k].grad.double().norm().item() for k in names])
    np.testing.assert_allclose(norms, g["head_grad_norms"], rtol=2e-3, atol=1e-5)


def test_partseg_oracle_at_the_finetune_batch_vs_reference():
    """BASELINE configs[4] at its batch (16 clouds of 1024 points on the configs[2] backbone; fixture: make_golden.py make_partseg_full,
    written by the imported reference): eval / train logits, the label-smoothed cross entropy and the gradient of THAT loss for every
    parameter (norms + strided samples)."""
    name = "c3"
    arch, a = _arch(name)
    g = Hh.golden(f"fullsize_partseg_{name}.npz")
    B = int(g["meta"][0])
    pts, start, cls, target = Hh.partseg_inputs(name, B, int(g["meta"][1]))
    lidx = Hh.PARTSEG_LAYERS[name]
    sd = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_partseg_{name}.json"), 100)
    with torch.no_grad():
        _close(O.partseg_forward(sd, pts, start, cls, arch, lidx, False)[:, :64], g["eval_logits"], 2e-4, 2e-4)
    names = json.load(open(os.path.join(Hh.GOLDEN_DIR, f"grad_names_partseg_{name}.json")))
    hp = {k: sd[k].clone().requires_grad_() for k in names}
    sd2 = dict(sd); sd2.update(hp)
    for k in list(sd2):
        if "cross_attn_1." in k:
            sd2[k] = sd2[k.replace("cross_attn_1.", "cross_attn_n.")]
    keep_all = torch.full((B * a["N"], 512), 0.5)               # the fixture ran the head's Dropout(0.5) at p = 0: keep * 2 == 1
    y = O.partseg_forward(sd2, pts, start, cls, arch, lidx, True, O.Masks("off"), {}, head_mask=keep_all)
    _close(y[:, :64], g["train_logits"], 5e-4, 5e-4)
    loss = torch.nn.functional.cross_entropy(y.reshape(-1, 50), target.reshape(-1), label_smoothing=0.2)
    assert abs(loss.item() - float(g["ce_loss"][0])) < 1e-5
    loss.backward()
    zero = lambda p: p.grad if p.grad is not None else torch.zeros_like(p)
    norms = np.array([zero(hp[k]).double().norm().item() for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=5e-3, atol=1e-5 * g["grad_norms"].max())
    smp = torch.cat([Hh.grad_sample(zero(hp[k])) for k in names]).double()
    ref = torch.from_numpy(g["grad_samples"]).double()
    assert float(smp @ ref / (smp.norm() * ref.norm())) > 1 - 1e-6


def _partseg_inputs(name):
    a = Hh.ARCHS[name]
    B = Hh.MODEL_BATCH[name]
    pts = Hh.synth_points(320, B, a["N"]); start = Hh.synth_start(320, B, a["N"])
    cls = torch.zeros(B, 16)
    cls[torch.arange(B), torch.arange(B) % 16] = 1.0
    target = torch.from_numpy((np.random.default_rng(321).random((B, a["N"])) * 50).astype(np.int64))
    return a, B, pts, start, cls, target


@pytest.mark.parametrize("name", ["tinyseg", "c3"])
def test_partseg_vs_reference(name):
    """oracle.partseg_forward / feature_propagation / three_nn vs CrossFormer_partseg + PointNetFeaturePropagation of the reference
    (partseg.py:345-470, utils.py:192-242; fixture: make_golden.py make_partseg).  BASELINE config 5 = the c3 architecture."""
    arch, _ = _arch(name)
    a, B, pts, start, cls, target = _partseg_inputs(name)
    lidx = Hh.PARTSEG_LAYERS[name]
    g = Hh.golden(f"partseg_{name}.npz")
    sd = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_partseg_{name}.json"), 100)
    with torch.no_grad():
        _close(O.partseg_forward(sd, pts, start, cls, arch, lidx, False)[:, :64], g["eval_logits"], 2e-4, 2e-4)
    names = json.load(open(os.path.join(Hh.GOLDEN_DIR, f"grad_names_partseg_{name}.json")))
    hp = {k: sd[k].clone().requires_grad_() for k in names}
    sd2 = dict(sd); sd2.update(hp)
    for k in list(sd2):
        if "cross_attn_1." in k:
            sd2[k] = sd2[k.replace("cross_attn_1.", "cross_attn_n.")]
    bufs = {}
    keep_all = torch.full((B * a["N"], 512), 0.5)               # the fixture ran the head's Dropout(0.5) at p = 0: keep * 2 == 1
    y = O.partseg_forward(sd2, pts, start, cls, arch, lidx, True, O.Masks("off"), bufs, head_mask=keep_all)
    _close(y[:, :64], g["train_logits"], 5e-4, 5e-4)
    loss = torch.nn.functional.cross_entropy(y.reshape(-1, 50), target.reshape(-1), label_smoothing=0.2)
    np.testing.assert_allclose(loss.item(), g["ce_loss"][0], rtol=1e-5)
    (y * Hh.synth_like(720, y.shape)).sum().backward()
    norms = np.array([hp[k].grad.double().norm().item() if hp[k].grad is not None else 0.0 for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=5e-3, atol=1e-4 * float(g["grad_norms"].max()))
    for k in ("bn1.running_mean", "bn1.running_var", "propagation.mlp_bns.1.running_var", "label_conv.1.running_var"):
        _close(bufs[k], g["buf." + k], 1e-4, 1e-5)
    # the 3-NN stage: indices and weight BITS
    nb, ct, _ = O.divide_patches(pts, O.fps_indices(pts, start, arch.G), arch.K)
    idx, w = O.three_nn(pts, ct)
    assert bool(g["nn_tie_free"])
    assert np.array_equal(idx.numpy(), g["nn_idx"])
    assert np.array_equal(w.numpy().view(np.uint32), g["nn_weight_bits"])


def test_augmentation_trans1_vs_reference():
    """oracle/augment.py trans_1 vs the reference's data_utils.py classes under the same numpy / torch seeds
    (datasets/data.py:16-25; fixture: make_golden.py make_augment).  Exact: same operations, same RNG draws."""
    from oracle import augment as A
    g = Hh.golden("augment_trans1.npz")
    for case in range(3):
        seed, n = (int(v) for v in g[f"meta{case}"])
        cloud = (np.random.default_rng(seed).random((n, 3)) * 2.0 - 1.0).astype(np.float32) * np.array([1.0, 0.5, 2.0], np.float32) + 0.3
        np.random.seed(seed); torch.manual_seed(seed)
        out = A.trans_1(cloud.copy())
        assert np.array_equal(out.numpy(), g[f"out{case}"]), case
    img = (np.random.default_rng(1).random((137, 137, 3)) * 255).astype(np.uint8)
    np.random.seed(3)
    y = A.image_transform(img, 224)
    assert y.shape == (3, 224, 224) and y.dtype == np.float32 and np.isfinite(y).all()


def test_two_cross_attention_layers_vs_reference():
    """Encoder with num_cross_attention_layers = 2 (partseg.py:297-300,331-334; fixture: make_golden.py make_ca2)."""
    arch, a = _arch("tiny")
    arch.n_ca = 2
    g = Hh.golden("model_tiny_ca2.npz")
    sd = Hh.synth_state_dict(Hh.load_keyshapes("keys_pc_tiny_ca2.json"), 100, alias_ca=False)
    B = Hh.MODEL_BATCH["tiny"]
    pts = Hh.synth_points(300, 2 * B, a["N"]); start = Hh.synth_start(300, 2 * B, a["N"])
    with torch.no_grad():
        f, bb = O.pc_forward(sd, pts, start, arch, False)
    _close(f, g["pc_eval_feats"], 1e-4, 1e-4); _close(bb, g["pc_eval_backbone"], 1e-4, 1e-4)


def _padmask_state(g, prefix, seed):
    """State dict of one pad_mask fixture module: the key shapes are those of its stored gradients (the modules hold no buffers)."""
    ks = [(k[len(prefix):], tuple(g[k].shape)) for k in g.files if k.startswith(prefix)]
    return Hh.synth_state_dict(ks, seed, alias_ca=False), [k for k, _ in ks]


def test_pad_mask_vs_reference():
    """Key padding mask (partseg.py:53-86, 111-116, 138-141, 314-335; fixture: make_golden.py make_padmask): a cross-attention layer,
    a self-attention and a whole Encoder with pad_mask -- ragged lengths, a scattered mask and a batch row whose keys are ALL padded
    (uniform attention) -- outputs and every gradient of a linear loss against the reference run."""
    c = Hh.padmask_inputs()
    g = Hh.golden("padmask.npz")
    arch = O.Arch(D=c["D"], H=c["H"], S=2, MR=2, atten_drop=0.0, mlp_drop=0.0)
    off = O.Masks("off")
    # (1) cross-attention layer
    sd, names = _padmask_state(g, "ca_g.", 910)
    sd = {k: v.clone().requires_grad_() for k, v in sd.items()}
    xq, xkv = c["xq"].clone().requires_grad_(), c["xkv"].clone().requires_grad_()
    y = O.ca_layer(sd, "", xq, xkv, arch, off, "ca", c["pad"])
    (y * c["R"]).sum().backward()
    _close(y.detach(), g["ca_out"], 1e-4, 1e-4); _close(xq.grad, g["ca_dxq"], 1e-4, 2e-3); _close(xkv.grad, g["ca_dxkv"], 1e-4, 2e-3)
    for k in names:
        _close(sd[k].grad, g["ca_g." + k], 1e-4, 2e-3)
    assert float(xkv.grad[2].abs().max()) > 0.0           # the all-padded row still feeds v (uniform attention) ...
    # (2) self-attention with a mask over its own tokens
    sd, names = _padmask_state(g, "sa_g.", 920)
    sd = {k: v.clone().requires_grad_() for k, v in sd.items()}
    x = c["xq"].clone().requires_grad_()
    n = torch.nn.functional.layer_norm(x, (c["D"],), sd["norm.weight"], sd["norm.bias"], 1e-5)
    y = O.mha(sd, "attention.", n, n, c["H"], 0.0, off, "sa", c["pad_self"])
    (y * c["R"]).sum().backward()
    _close(y.detach(), g["sa_out"], 1e-4, 1e-4); _close(x.grad, g["sa_dx"], 1e-4, 2e-3)
    for k in names:
        _close(sd[k].grad, g["sa_g." + k], 1e-4, 2e-3)
    # (3) Encoder.forward(..., pad_mask): the mask reaches the cross-attention layer only
    sd = Hh.synth_state_dict(Hh.load_keyshapes("keys_enc_padmask.json"), 930)
    sd = {k: (v.clone().requires_grad_() if "cross_attn_1." not in k else v) for k, v in sd.items()}
    for k in list(sd):
        if "cross_attn_1." in k:
            sd[k] = sd[k.replace("cross_attn_1.", "cross_attn_n.")]
    tok, kv = c["xq"].clone().requires_grad_(), c["xkv"].clone().requires_grad_()
    y, _ = O.encoder(sd, "", tok, c["pos"], kv, arch, off, (), c["pad"])
    (y * c["R"]).sum().backward()
    _close(y.detach(), g["enc_out"], 1e-4, 1e-4); _close(tok.grad, g["enc_dtok"], 1e-4, 2e-3); _close(kv.grad, g["enc_dkv"], 1e-4, 2e-3)
    for k in g.files:
        if k.startswith("enc_g."):
            _close(sd[k[6:]].grad, g[k], 1e-4, 2e-3)


def test_ntxent_restatement_vs_the_published_formula_and_closed_forms():
    """NT-Xent is PARITY-UNPINNED (lightly==1.1.21 is absent from the image and from /root/reference: SURVEY 8a row 17).  What can be
    checked without it: (1) the oracle's vectorised restatement (mask the diagonal, cross entropy over 2b - 1 logits) against the SimCLR
    paper's equation written out sample by sample in float64 -- l(i, j) = -log(exp(s_ij / T) / sum_{k != i} exp(s_ik / T)), mean over the
    2b ordered positive pairs, s = cosine similarity; (2) closed forms: b mutually orthogonal pairs with identical views give
    -log(e^{1/T} / (e^{1/T} + 2b - 2)); all 2b embeddings identical give log(2b - 1) (every negative ties with the positive)."""
    import math
    T = 0.1
    g = torch.Generator().manual_seed(5)
    for b, d in ((2, 8), (5, 16), (64, 256)):
        z0, z1 = torch.randn(b, d, generator=g), torch.randn(b, d, generator=g) * 3.0 + 0.5
        z = torch.cat([z0, z1]).double()
        z = z / z.norm(dim=1, keepdim=True)
        s = (z @ z.t()) / T
        tot = 0.0
        for i in range(2 * b):
            j = (i + b) % (2 * b)
            den = sum(math.exp(float(s[i, k])) for k in range(2 * b) if k != i)
            tot += -math.log(math.exp(float(s[i, j])) / den)
        assert abs(O.ntxent(z0, z1, T).item() - tot / (2 * b)) < 2e-5 * max(1.0, abs(tot / (2 * b)))
    for b in (2, 7, 32):
        e = torch.eye(b, 2 * b)                                   # b orthonormal directions, the two views identical
        want = -math.log(math.exp(1 / T) / (math.exp(1 / T) + 2 * b - 2))
        assert abs(O.ntxent(e, e.clone(), T).item() - want) < 1e-5 * max(1.0, want)
        same = torch.ones(b, 4)
        assert abs(O.ntxent(same, same.clone(), T).item() - math.log(2 * b - 1)) < 1e-5
