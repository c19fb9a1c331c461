"""Regenerates tests/golden/* by IMPORTING THE REFERENCE (build container only).

    python tests/golden/make_golden.py            # needs /root/reference

The reference never travels to the GPU box; these fixtures (inputs are re-derived
from seeds by tests/helpers.py, only expected outputs are stored) are what pins the
oracle, and through it the HIP path.

Import shim: fairscale.nn.checkpoint_wrapper and timm DropPath are
import-time-only dependencies of partseg.py (checkpointing is never enabled and
max_dpr=0 -> nn.Identity), so identity stand-ins are registered before the import.

kNN ordering policy: torch.topk(sorted=False) order is unspecified and feeds the
utils.py:36 axis quirk, so for every fixture downstream of knn_point the reference
is run with knn_point replaced by its canonical-order equivalent (stable ascending
sort = ascending distance, ties -> lower index).  The raw reference output is kept
as a sorted SET in knn_*.npz.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("VPF_REFERENCE", "/root/reference")


def install_shim():
    fs = types.ModuleType("fairscale"); fsnn = types.ModuleType("fairscale.nn")
    fsnn.checkpoint_wrapper = lambda m, *a, **k: m
    fs.nn = fsnn
    sys.modules["fairscale"] = fs; sys.modules["fairscale.nn"] = fsnn
    timm = types.ModuleType("timm"); tm = types.ModuleType("timm.models"); tl = types.ModuleType("timm.models.layers")

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
        def forward(self, x):
            return x
    tl.DropPath = DropPath; timm.models = tm; tm.layers = tl
    sys.modules["timm"] = timm; sys.modules["timm.models"] = tm; sys.modules["timm.models.layers"] = tl
    sys.path.insert(0, REF)


install_shim()
from vipformer.model.pointcloud import PointCloudInputAdapter, CrossFormer_pc_mp, CrossFormer_img_mp  # noqa: E402
from vipformer.model.pointcloud import utils as RU  # noqa: E402
from vipformer.model.pointcloud import partseg as RP  # noqa: E402

from tests import helpers as Hh  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

_orig_knn = RU.knn_point


def canonical_knn(nsample, xyz, new_xyz):
    d = RU.square_distance(new_xyz, xyz)
    return torch.sort(d, dim=-1, stable=True)[1][:, :, :nsample]


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print("wrote", name, {k: tuple(np.asarray(v.detach() if torch.is_tensor(v) else v).shape) for k, v in arrs.items()})


def fps_with_start(pts, G, start):
    """Run the reference FPS with its randint (utils.py:71) forced to ``start``."""
    real = torch.randint
    torch.randint = lambda *a, **k: start.clone()
    try:
        return RU.farthest_point_sample(pts, G)
    finally:
        torch.randint = real


class forced_start:
    def __init__(self, start):
        self.start = start
    def __enter__(self):
        self.real = torch.randint
        torch.randint = lambda *a, **k: self.start.clone()
    def __exit__(self, *a):
        torch.randint = self.real


def build_ref(a, drops=(0.1, 0.5)):
    ad = PointCloudInputAdapter((a["N"], 3), a["D"])
    pc = CrossFormer_pc_mp(ad, a["G"], a["D"], a["K"], 1, a["H"], a["S"], a["H"], a["MR"], 0.0, drops[0], drops[1], True)
    im = CrossFormer_img_mp(a["img"], a["img"], a["patch"], a["D"], 1, a["H"], a["S"], a["H"], a["MR"], 0.0, drops[0], drops[1], True)
    return pc, im


def keyshapes(m):
    return [(k, list(v.shape)) for k, v in m.state_dict().items()]


def grad_summary(model):
    """per-parameter L2 norm + first 8 elements (keeps fixtures small)."""
    names, norms, heads = [], [], []
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        names.append(k); norms.append(g.double().norm().item())
        h = g.reshape(-1)[:8]
        heads.append(torch.cat([h, torch.zeros(8 - h.numel())]).numpy())
    return names, np.array(norms), np.stack(heads)


def make_keys(names):
    """state-dict key lists + shapes of the reference's two models for architectures added after round 1 (keys_*_<name>.json)."""
    for name in names:
        pc, im = build_ref(Hh.ARCHS[name])
        json.dump(keyshapes(pc), open(os.path.join(HERE, f"keys_pc_{name}.json"), "w"))
        json.dump(keyshapes(im), open(os.path.join(HERE, f"keys_img_{name}.json"), "w"))
        cpath = os.path.join(HERE, "param_counts.json")
        counts = json.load(open(cpath))
        counts[name] = dict(pc_params=sum(p.numel() for p in pc.parameters()), img_params=sum(p.numel() for p in im.parameters()),
                            pc_state=len(pc.state_dict()), img_state=len(im.state_dict()),
                            pc_named=[k for k, _ in pc.named_parameters()], img_named=[k for k, _ in im.named_parameters()])
        json.dump(counts, open(cpath, "w"))
        print("keys", name, counts[name]["pc_params"], counts[name]["img_params"])


def main(only_models=None):
    torch.set_num_threads(8)
    if only_models:
        make_keys([n for n in only_models if not os.path.exists(os.path.join(HERE, f"keys_pc_{n}.json"))])
        return make_models(only_models)
    # ------------------------------------------------------------ key lists / param counts
    counts = {}
    for name in ("c1", "c3", "c4", "tiny", "tiny2"):
        a = Hh.ARCHS[name]
        pc, im = build_ref(a)
        json.dump(keyshapes(pc), open(os.path.join(HERE, f"keys_pc_{name}.json"), "w"))
        json.dump(keyshapes(im), open(os.path.join(HERE, f"keys_img_{name}.json"), "w"))
        counts[name] = dict(pc_params=sum(p.numel() for p in pc.parameters()),
                            img_params=sum(p.numel() for p in im.parameters()),
                            pc_state=len(pc.state_dict()), img_state=len(im.state_dict()),
                            pc_named=[k for k, _ in pc.named_parameters()],
                            img_named=[k for k, _ in im.named_parameters()])
    json.dump(counts, open(os.path.join(HERE, "param_counts.json"), "w"))
    # default-initialisation fingerprint: torch.manual_seed(1) (parser.py:17) then build_model order (utils.py:115-149)
    init = {}
    for name in ("tiny", "c1"):
        a = Hh.ARCHS[name]
        torch.manual_seed(1)
        pc, im = build_ref(a)
        init[name] = dict(pc={k: float(v.double().sum()) for k, v in pc.state_dict().items()},
                          img={k: float(v.double().sum()) for k, v in im.state_dict().items()})
    json.dump(init, open(os.path.join(HERE, "init_checksums.json"), "w"))
    print({k: (v["pc_params"], v["img_params"], v["pc_state"], v["img_state"]) for k, v in counts.items()})

    # ------------------------------------------------------------ FPS / sqdist / kNN / divide_patches
    cases = [("u1024", 11, 4, 1024, 3, 96, 32, "uniform"), ("u2048", 12, 2, 2048, 3, 128, 32, "uniform"),
             ("d1024", 13, 4, 1024, 3, 96, 32, "dups"), ("g1024", 14, 3, 1024, 3, 96, 32, "grid"),
             ("c6_1000", 15, 3, 1000, 6, 7, 8, "uniform"), ("u256", 16, 4, 256, 3, 16, 8, "uniform"),
             ("u512", 17, 4, 512, 3, 32, 16, "uniform")]
    for (tag, seed, B, N, C, G, K, mode) in cases:
        pts = Hh.synth_points(seed, B, N, C, mode)
        start = Hh.synth_start(seed, B, N)
        idx = fps_with_start(pts, G, start)
        assert (idx[:, 0] == start).all()
        centers = RU.index_points(pts, idx)
        d = RU.square_distance(centers[:, :, :3], pts[:, :, :3])
        raw = _orig_knn(K, pts[:, :, :3], centers[:, :, :3])
        can = canonical_knn(K, pts[:, :, :3], centers[:, :, :3])
        srt = torch.sort(d, dim=-1, stable=True)[0]
        tie_free = bool((srt[:, :, K - 1] != srt[:, :, K]).all())
        RU.knn_point = canonical_knn
        with forced_start(start):
            nb, ct = RU.divide_patches(pts, G, K)
        RU.knn_point = _orig_knn
        small = B * G * N <= 4 * 16 * 512
        save(f"preproc_{tag}.npz", fps_idx=idx, knn_canonical=can, knn_raw_sorted=torch.sort(raw, -1)[0],
             tie_free=np.array(tie_free), neighbors=nb, centers=ct,
             sqdist_bits=(d.numpy().view(np.uint32) if small else d[:, :4, :].contiguous().numpy().view(np.uint32)),
             knn_dist=torch.gather(d, 2, can), meta=np.array([seed, B, N, C, G, K]))

    make_keys(Hh.REF_ARCHS)
    make_models(("tiny", "tiny2", "c1", "c3", "c4") + Hh.REF_ARCHS)
    print("done")


SLICED = Hh.FULLSIZE      # full-size architectures: stage outputs / big gradients are stored as slices


def make_models(names):
    # ------------------------------------------------------------ stage + model goldens
    RU.knn_point = canonical_knn
    RP.divide_patches.__globals__["knn_point"] = canonical_knn
    for name in names:
        a = Hh.ARCHS[name]
        B = 2
        pc, im = build_ref(a, drops=(0.0, 0.0))
        pc.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pc_{name}.json"), 100))
        im.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes(f"keys_img_{name}.json"), 200))
        pts = Hh.synth_points(300, 2 * B, a["N"], 3, "uniform")
        start = Hh.synth_start(300, 2 * B, a["N"])
        imgs = Hh.synth_images(400, B, a["img"], a["img"])
        out = {}
        # ---- stage: Group2Emb (train + eval), with weight grads
        with forced_start(start):
            nb, ct = RU.divide_patches(pts, a["G"], a["K"])
        g2e = pc.group2emb
        g2e.train(); g2e.zero_grad()
        y = g2e(nb)
        R = Hh.synth_like(500, y.shape)
        (y * R).sum().backward()
        out["g2e_train"] = y
        out["g2e_rm1"] = g2e.first_conv[1].running_mean.clone(); out["g2e_rv1"] = g2e.first_conv[1].running_var.clone()
        out["g2e_rm2"] = g2e.second_conv[1].running_mean.clone(); out["g2e_rv2"] = g2e.second_conv[1].running_var.clone()
        for k, p in g2e.named_parameters():
            out["g2e_grad." + k] = p.grad.clone()
        pc.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pc_{name}.json"), 100))
        g2e.eval()
        out["g2e_eval"] = g2e(nb)
        # ---- stage: adapter, pos
        out["adapter"] = pc.input_adapter(pts)[:, :32].clone()
        out["pos"] = pc.position_emb(ct)
        # ---- stage: one SA layer + the CA layer, fwd + input grads (dropout 0)
        pc.train()
        enc = pc.encoder
        x = Hh.synth_like(600, (2 * B, a["G"], a["D"])).requires_grad_()
        kv = Hh.synth_like(601, (2 * B, a["N"], a["D"])).requires_grad_()
        pc.zero_grad()
        yca = enc.cross_attn_1(x, kv, None)
        Rl = Hh.synth_like(602, yca.shape)
        (yca * Rl).sum().backward()
        out["ca_out"] = yca; out["ca_dx"] = x.grad.clone(); out["ca_dkv"] = kv.grad[:, :32].clone()
        for k, p in enc.cross_attn_n.named_parameters():
            out["ca_grad." + k] = p.grad.clone()
        x2 = Hh.synth_like(603, (2 * B, a["G"], a["D"])).requires_grad_()
        pc.zero_grad()
        ysa = enc.sa_layers[0](x2)
        (ysa * Rl).sum().backward()
        out["sa_out"] = ysa; out["sa_dx"] = x2.grad.clone()
        for k, p in enc.sa_layers[0].named_parameters():
            out["sa_grad." + k] = p.grad.clone()
        if name in SLICED:   # keep the full-size fixtures small: stage outputs only as slices
            for k in list(out):
                if k.startswith(("ca_grad.", "sa_grad.", "g2e_grad.")) and out[k].numel() > 4096:
                    out[k] = out[k].reshape(-1)[:4096].clone()
            out["g2e_train"] = out["g2e_train"][:, :8].clone(); out["g2e_eval"] = out["g2e_eval"][:, :8].clone()
            out["ca_out"] = out["ca_out"][:, :8].clone(); out["sa_out"] = out["sa_out"][:, :8].clone()
            out["ca_dx"] = out["ca_dx"][:, :8].clone(); out["sa_dx"] = out["sa_dx"][:, :8].clone()
            out["pos"] = out["pos"][:, :8].clone()
        save(f"stages_{name}.npz", **out)

        # ---- full models: eval, train (dropout 0) + grads.  More pairs than the stage fixtures: the projection
        #      head's BatchNorm normalises over the batch, and 4 samples make it needlessly ill-conditioned.
        B = Hh.MODEL_BATCH[name]
        pts = Hh.synth_points(300, 2 * B, a["N"], 3, "uniform")
        start = Hh.synth_start(300, 2 * B, a["N"])
        imgs = Hh.synth_images(400, B, a["img"], a["img"])
        res = {}
        pc.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pc_{name}.json"), 100))
        pc.eval(); im.eval()
        with torch.no_grad(), forced_start(start):
            f, bb = pc(pts)
        res["pc_eval_feats"], res["pc_eval_backbone"] = f, bb
        with torch.no_grad():
            f, bb = im(imgs)
        res["img_eval_feats"], res["img_eval_backbone"] = f, bb
        pc.train(); im.train(); pc.zero_grad(); im.zero_grad()
        with forced_start(start):
            f, bb = pc(pts)
        fi, bbi = im(imgs)
        res["pc_train_feats"], res["pc_train_backbone"] = f, bb
        res["img_train_feats"], res["img_train_backbone"] = fi, bbi
        # the pretrain.py:189-207 loss VALUE with the (unpinned) NT-Xent restatement
        f1, f2 = f[:B], f[B:]
        l_im = O.ntxent(f1, f2); l_cm = O.ntxent((f1 + f2) / 2, fi)
        res["loss"] = np.array([(l_im + l_cm).item(), l_im.item(), l_cm.item()])
        # gradients of a loss that is LINEAR in the backbone features (well-conditioned: the projection head's
        # BatchNorm over a handful of samples and the temperature-0.1 softmax amplify any forward difference,
        # which would make a gradient comparison measure the forward error instead of the backward kernels)
        (bb * Hh.synth_like(700, bb.shape)).sum().backward()
        (bbi * Hh.synth_like(701, bbi.shape)).sum().backward()
        n, norms, heads = grad_summary(pc)
        res["pc_grad_norms"], res["pc_grad_heads"] = norms, heads
        n2, norms2, heads2 = grad_summary(im)
        res["img_grad_norms"], res["img_grad_heads"] = norms2, heads2
        for k in ("latent_head.0.running_mean", "latent_head.0.running_var", "group2emb.first_conv.1.running_var"):
            res["pc_buf." + k] = pc.state_dict()[k].clone()
        save(f"model_{name}.npz", **res)
        json.dump(dict(pc=n, img=n2), open(os.path.join(HERE, f"grad_names_{name}.json"), "w"))

        # ---- dropout placement pin (oracle-only test): train mode, real p, torch RNG stream
        if name not in ("c1", "c3", "c4"):
            pcd, imd = build_ref(a, drops=(0.1, 0.5))
            pcd.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pc_{name}.json"), 100))
            imd.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes(f"keys_img_{name}.json"), 200))
            pcd.train(); imd.train()
            torch.manual_seed(77)
            with forced_start(start):
                fd, bd = pcd(pts)
            torch.manual_seed(78)
            fid, bid = imd(imgs)
            save(f"dropout_{name}.npz", pc_feats=fd, pc_backbone=bd, img_feats=fid, img_backbone=bid)

def make_fullsize(names=("c1", "c3", "c4")):
    """BASELINE configs[1..3] at their PER-GPU BATCH (64 / 32 / 16 pairs: tests/helpers.py FULL_BATCH) through the imported reference
    (full size pinned to the reference itself, not to the restatement run live on the GPU box) -> fullsize_<name>.npz:
      * eval-mode features + backbone of both models;
      * train mode with every dropout at 0 (BatchNorm on batch statistics): features, backbone, the pre-training loss
        (pretrain.py:189-207; the NT-Xent VALUE comes from the restatement -- lightly is absent, row 17 stays unpinned -- but everything
        in front of it and the whole backward pass through both models is the reference's);
      * gradients of (a) the loss linear in the backbone features and (b) the pre-training loss, for every parameter: L2 norm and a
        strided sample of <= 512 elements (helpers.grad_sample: the same indices are taken from the HIP gradients by the test), so the
        fixture stays at a few hundred KB where the full gradients are 2 x 33 - 130 MB."""
    RU.knn_point = canonical_knn
    RP.divide_patches.__globals__["knn_point"] = canonical_knn
    torch.set_num_threads(8)
    for name in names:
        a = Hh.ARCHS[name]
        B = Hh.FULL_BATCH[name]
        pc, im = build_ref(a, drops=(0.0, 0.0))
        pc.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pc_{name}.json"), 100))
        im.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes(f"keys_img_{name}.json"), 200))
        pts = Hh.synth_points(930, 2 * B, a["N"], 3, "uniform")
        start = Hh.synth_start(930, 2 * B, a["N"])
        imgs = Hh.synth_images(931, B, a["img"], a["img"])
        res = {"meta": np.array([B, 930, 931])}
        pc.eval(); im.eval()
        with torch.no_grad(), forced_start(start):
            f, bb = pc(pts)
            fi, bbi = im(imgs)
        res.update(pc_eval_feats=f, pc_eval_backbone=bb, img_eval_feats=fi, img_eval_backbone=bbi)
        pc.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pc_{name}.json"), 100))     # (nothing moved in eval mode; belt and braces)
        pc.train(); im.train(); pc.zero_grad(); im.zero_grad()
        with forced_start(start):
            f, bb = pc(pts)
        fi, bbi = im(imgs)
        res.update(pc_train_feats=f, pc_train_backbone=bb, img_train_feats=fi, img_train_backbone=bbi)
        f1, f2 = f[:B], f[B:]
        l_im = O.ntxent(f1, f2); l_cm = O.ntxent((f1 + f2) / 2, fi)
        total = l_im + 1.0 * l_cm                                                                  # pretrain.py:207
        res["loss"] = np.array([total.item(), l_im.item(), l_cm.item()])
        names_pc = [k for k, _ in pc.named_parameters()]; names_im = [k for k, _ in im.named_parameters()]
        assert dict(pc=names_pc, img=names_im) == json.load(open(os.path.join(HERE, f"grad_names_{name}.json")))
        for tag, loss in (("lin", (bb * Hh.synth_like(700, bb.shape)).sum() + (bbi * Hh.synth_like(701, bbi.shape)).sum()), ("ntx", total)):
            pc.zero_grad(); im.zero_grad()
            loss.backward(retain_graph=(tag == "lin"))
            for which, model in (("pc", pc), ("img", im)):
                norms, samples = [], []
                for k, p_ in model.named_parameters():
                    g = p_.grad if p_.grad is not None else torch.zeros_like(p_)
                    norms.append(g.double().norm().item())
                    samples.append(Hh.grad_sample(g))
                res[f"{which}_{tag}_norms"] = np.array(norms)
                res[f"{which}_{tag}_samples"] = torch.cat(samples)
        for k in ("latent_head.0.running_mean", "latent_head.0.running_var", "group2emb.first_conv.1.running_var"):
            res["pc_buf." + k] = pc.state_dict()[k].clone()
        save(f"fullsize_{name}.npz", **res)
        print("fullsize", name, B, "pairs: loss", res["loss"], os.path.getsize(os.path.join(HERE, f"fullsize_{name}.npz")), "bytes", flush=True)
    RU.knn_point = _orig_knn


def make_ft(names=("tiny", "c1")):
    """CrossFormer_pc_mp_ft (partseg.py:553-605, the ModelNet fine-tuning classifier: the pre-training backbone + a 3-block
    BatchNorm-ReLU-Linear head): eval / train logits (dropout 0), the gradients of a loss linear in the logits with respect to
    the head, and the first BatchNorm's running statistics after one training-mode forward."""
    NCLS = 40
    for name in names:
        a = Hh.ARCHS[name]
        torch.manual_seed(0)
        ad = PointCloudInputAdapter((a["N"], 3), a["D"])
        ft = RP.CrossFormer_pc_mp_ft(ad, a["G"], a["D"], a["K"], 1, a["H"], a["S"], a["H"], a["MR"], 0.0, 0.0, 0.0, True, NCLS)
        json.dump(keyshapes(ft), open(os.path.join(HERE, f"keys_pcft_{name}.json"), "w"))
        ft.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes(f"keys_pcft_{name}.json"), 100))
        B = Hh.MODEL_BATCH[name]
        pts = Hh.synth_points(300, 2 * B, a["N"], 3, "uniform")
        start = Hh.synth_start(300, 2 * B, a["N"])
        res = {}
        RU.knn_point = canonical_knn
        ft.eval()
        with torch.no_grad(), forced_start(start):
            res["eval_logits"] = ft(pts)
        ft.train(); ft.zero_grad()
        with forced_start(start):
            y = ft(pts)
        res["train_logits"] = y
        (y * Hh.synth_like(710, y.shape)).sum().backward()
        hp = [(k, p) for k, p in ft.named_parameters() if k.startswith("finetune_head.")]
        res["head_grad_norms"] = np.array([p.grad.double().norm().item() for _, p in hp])
        json.dump([k for k, _ in hp], open(os.path.join(HERE, f"grad_names_pcft_{name}.json"), "w"))
        for k in ("finetune_head.0.running_mean", "finetune_head.0.running_var", "finetune_head.6.running_var"):
            res["buf." + k] = ft.state_dict()[k].clone()
        RU.knn_point = _orig_knn
        save(f"modelft_{name}.npz", **res)
        print("ft fixture", name, {k: tuple(np.asarray(v.detach() if torch.is_tensor(v) else v).shape) for k, v in res.items()})


def make_partseg(names=("tinyseg", "c3")):
    """CrossFormer_partseg (partseg.py:345-470; BASELINE config 5 is the c3 architecture): key list, eval / train logits with every
    encoder dropout at 0 (the head's Dropout(0.5) is only active in train mode, so the train-mode fixture is taken with the head
    dropout patched to 0 as well and checked for the BatchNorm batch statistics), gradients of a loss linear in the logits, the
    3-NN indices / weights of PointNetFeaturePropagation, and the label-smoothed cross entropy of ft_partseg.py:128."""
    NPART = 50
    for name in names:
        a = Hh.ARCHS[name]
        lidx = Hh.PARTSEG_LAYERS[name]
        torch.manual_seed(0)
        ad = PointCloudInputAdapter((a["N"], 3), a["D"])
        m = RP.CrossFormer_partseg(ad, a["G"], a["D"], a["K"], 1, a["H"], a["S"], a["H"], a["MR"], 0.0, 0.0, 0.0, lidx, NPART)
        json.dump(keyshapes(m), open(os.path.join(HERE, f"keys_partseg_{name}.json"), "w"))
        m.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes(f"keys_partseg_{name}.json"), 100))
        B = Hh.MODEL_BATCH[name]
        pts = Hh.synth_points(320, B, a["N"], 3, "uniform")
        start = Hh.synth_start(320, B, a["N"])
        cls = torch.zeros(B, 16)
        cls[torch.arange(B), torch.arange(B) % 16] = 1.0
        target = torch.from_numpy((np.random.default_rng(321).random((B, a["N"])) * NPART).astype(np.int64))
        res = {}
        RU.knn_point = canonical_knn
        m.eval()
        with torch.no_grad(), forced_start(start):
            res["eval_logits"] = m(pts, cls)[:, :64].clone()
        m.train(); m.zero_grad()
        m.dp1.p = 0.0
        with forced_start(start):
            y = m(pts, cls)
        res["train_logits"] = y[:, :64].clone()
        loss = torch.nn.CrossEntropyLoss(label_smoothing=0.2)(y.reshape(-1, NPART), target.reshape(-1))
        res["ce_loss"] = np.array([loss.item()])
        (y * Hh.synth_like(720, y.shape)).sum().backward()
        names_ = [k for k, p in m.named_parameters() if p.grad is not None]
        res["grad_norms"] = np.array([dict(m.named_parameters())[k].grad.double().norm().item() for k in names_])
        json.dump(names_, open(os.path.join(HERE, f"grad_names_partseg_{name}.json"), "w"))
        for k in ("bn1.running_mean", "bn1.running_var", "propagation.mlp_bns.1.running_var", "label_conv.1.running_var"):
            res["buf." + k] = m.state_dict()[k].clone()
        # the 3-NN stage on its own
        with forced_start(start):
            _, ct = RU.divide_patches(pts, a["G"], a["K"])
        d = RU.square_distance(pts, ct)
        dist, idx = torch.sort(d, dim=-1, stable=True)
        recip = 1.0 / (dist[:, :, :3] + 1e-8)
        res["nn_idx"] = idx[:, :, :3].clone()
        res["nn_weight_bits"] = (recip / recip.sum(dim=2, keepdim=True)).numpy().view(np.uint32)
        res["nn_tie_free"] = np.array(bool((dist[:, :, 2] != dist[:, :, 3]).all() and (dist[:, :, 0] != dist[:, :, 1]).all()
                                           and (dist[:, :, 1] != dist[:, :, 2]).all()))
        RU.knn_point = _orig_knn
        save(f"partseg_{name}.npz", **res)


def make_partseg_full(name="c3", B=16):
    """BASELINE configs[4] at ITS batch: CrossFormer_partseg on the configs[2] backbone, 16 clouds of 1024 points (parser.py's fine-tune
    defaults; bench.py --arch c5) through the imported reference -> fullsize_partseg_<name>.npz: eval / train logits (slices; the head's
    Dropout(0.5) patched to 0 as in make_partseg), the label-smoothed cross entropy of ft_partseg.py:128, and the gradient of THAT loss for
    every parameter (norm + strided sample, helpers.grad_sample) -- what one fine-tune step backpropagates."""
    NPART = 50
    a = Hh.ARCHS[name]
    lidx = Hh.PARTSEG_LAYERS[name]
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ad = PointCloudInputAdapter((a["N"], 3), a["D"])
    m = RP.CrossFormer_partseg(ad, a["G"], a["D"], a["K"], 1, a["H"], a["S"], a["H"], a["MR"], 0.0, 0.0, 0.0, lidx, NPART)
    m.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes(f"keys_partseg_{name}.json"), 100))
    pts, start, cls, target = Hh.partseg_inputs(name, B, 940)
    res = {"meta": np.array([B, 940])}
    RU.knn_point = canonical_knn
    m.eval()
    with torch.no_grad(), forced_start(start):
        res["eval_logits"] = m(pts, cls)[:, :64].clone()
    m.train(); m.zero_grad()
    m.dp1.p = 0.0
    with forced_start(start):
        y = m(pts, cls)
    res["train_logits"] = y[:, :64].clone()
    loss = torch.nn.CrossEntropyLoss(label_smoothing=0.2)(y.reshape(-1, NPART), target.reshape(-1))
    res["ce_loss"] = np.array([loss.item()])
    loss.backward()
    names_ = json.load(open(os.path.join(HERE, f"grad_names_partseg_{name}.json")))
    params = dict(m.named_parameters())
    res["grad_norms"] = np.array([params[k].grad.double().norm().item() for k in names_])
    res["grad_samples"] = torch.cat([Hh.grad_sample(params[k].grad) for k in names_])
    for k in ("bn1.running_mean", "bn1.running_var", "propagation.mlp_bns.1.running_var", "label_conv.1.running_var"):
        res["buf." + k] = m.state_dict()[k].clone()
    RU.knn_point = _orig_knn
    save(f"fullsize_partseg_{name}.npz", **res)


def make_augment():
    """datasets/data.py:16-25 (trans_1) run with the reference's own data_utils.py classes (loaded by file path: the datasets
    package itself imports h5py / torchvision) under np.random.seed / torch.manual_seed -> augment_trans1.npz."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_data_utils", os.path.join(REF, "datasets", "data_utils.py"))
    du = importlib.util.module_from_spec(spec); spec.loader.exec_module(du)
    steps = [du.PointcloudToTensor(), du.PointcloudNormalize(), du.PointcloudScale(lo=0.5, hi=2, p=1), du.PointcloudRotate(),
             du.PointcloudTranslate(0.5, p=1), du.PointcloudJitter(p=1), du.PointcloudRandomInputDropout(p=1)]
    outs = {}
    for case, (seed, n) in enumerate(((11, 1024), (12, 2048), (13, 300))):
        cloud = (np.random.default_rng(seed).random((n, 3)) * 2.0 - 1.0).astype(np.float32) * np.array([1.0, 0.5, 2.0], np.float32) + 0.3
        np.random.seed(seed); torch.manual_seed(seed)
        x = cloud.copy()
        for t in steps:
            x = t(x)
        outs[f"out{case}"] = x.numpy().copy()
        outs[f"meta{case}"] = np.array([seed, n])
    save("augment_trans1.npz", **outs)


def make_ckpt():
    """A pc_model_best.pth exactly as pretrain.py:283-285 writes it -- torch.save(module.state_dict()) of the REFERENCE's
    CrossFormer_pc_mp (tiny architecture, synthetic weights, seed 100) -> ckpt_pc_tiny.pth (a data file: tensors + key order)."""
    a = Hh.ARCHS["tiny"]
    pc, _ = build_ref(a)
    pc.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes("keys_pc_tiny.json"), 100))
    torch.save(pc.state_dict(), os.path.join(HERE, "ckpt_pc_tiny.pth"))
    print("wrote ckpt_pc_tiny.pth", os.path.getsize(os.path.join(HERE, "ckpt_pc_tiny.pth")))


def make_ca2():
    """Encoder with num_cross_attention_layers = 2 (partseg.py:297-300,331-334: a separate cross_attn_1, cross_attn_n re-applied in front
    of the first self-attention layer): key list + eval-mode features of the tiny architecture -> model_tiny_ca2.npz."""
    a = Hh.ARCHS["tiny"]
    torch.manual_seed(0)
    ad = PointCloudInputAdapter((a["N"], 3), a["D"])
    pc = CrossFormer_pc_mp(ad, a["G"], a["D"], a["K"], 2, a["H"], a["S"], a["H"], a["MR"], 0.0, 0.0, 0.0, True)
    json.dump(keyshapes(pc), open(os.path.join(HERE, "keys_pc_tiny_ca2.json"), "w"))
    pc.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes("keys_pc_tiny_ca2.json"), 100, alias_ca=False))
    B = Hh.MODEL_BATCH["tiny"]
    pts = Hh.synth_points(300, 2 * B, a["N"], 3, "uniform")
    start = Hh.synth_start(300, 2 * B, a["N"])
    RU.knn_point = canonical_knn
    pc.eval()
    with torch.no_grad(), forced_start(start):
        f, bb = pc(pts)
    pc.train(); pc.zero_grad()
    with forced_start(start):
        f2, bb2 = pc(pts)
    (bb2 * Hh.synth_like(700, bb2.shape)).sum().backward()
    n, norms, heads = grad_summary(pc)
    RU.knn_point = _orig_knn
    json.dump(n, open(os.path.join(HERE, "grad_names_tiny_ca2.json"), "w"))
    save("model_tiny_ca2.npz", pc_eval_feats=f, pc_eval_backbone=bb, pc_train_backbone=bb2, pc_grad_norms=norms)


def padmask_case():
    """Inputs of the pad_mask fixture (shared with the tests through tests/helpers.py: only expected outputs are stored).
    Ragged lengths (40 queries, 70 keys: neither a multiple of 32); batch row 0 pads its last 20 keys, row 1 a scattered half,
    row 2 EVERY key (softmax of equal scores: uniform attention, no gradient to q / k)."""
    return Hh.padmask_inputs()


def make_padmask():
    """Key padding mask of MultiHeadAttention (partseg.py:53-86) through the reference's CrossAttentionLayer, SelfAttention and Encoder
    -> padmask.npz: outputs and every gradient for a linear loss, eval mode (dropout off)."""
    c = padmask_case()
    D, H = c["D"], c["H"]
    out = {}
    # (1) one cross-attention layer (LayerNorms, MHA with the mask, residual, MLP)
    torch.manual_seed(0)
    ca = RP.CrossAttentionLayer(H, D, D, D, widening_factor=2)
    ca.load_state_dict(Hh.synth_state_dict(keyshapes(ca), 910, alias_ca=False))
    ca.eval()
    xq = c["xq"].clone().requires_grad_(); xkv = c["xkv"].clone().requires_grad_()
    y = ca(xq, xkv, c["pad"])
    (y * c["R"]).sum().backward()
    out.update(ca_out=y.detach(), ca_dxq=xq.grad, ca_dxkv=xkv.grad)
    for k, v in ca.named_parameters():
        out["ca_g." + k] = v.grad
    # (2) self-attention with a mask over its own tokens (SelfAttention.forward(x, pad_mask), partseg.py:138-141)
    sa = RP.SelfAttention(H, D)
    sa.load_state_dict(Hh.synth_state_dict(keyshapes(sa), 920, alias_ca=False))
    sa.eval()
    x = c["xq"].clone().requires_grad_()
    y = sa(x, c["pad_self"])
    (y * c["R"]).sum().backward()
    out.update(sa_out=y.detach(), sa_dx=x.grad)
    for k, v in sa.named_parameters():
        out["sa_g." + k] = v.grad
    # (3) Encoder.forward(group_embs, pos_embs, pts_embs, pad_mask=...) (partseg.py:314-342), 2 self-attention layers
    enc = RP.Encoder(num_latent_channels=D, num_cross_attention_heads=H, cross_attention_widening_factor=2, num_self_attention_layers=2,
                     num_self_attention_heads=H, self_attention_widening_factor=2, dpr_list=[0.0, 0.0], modal_prior=True)
    json.dump(keyshapes(enc), open(os.path.join(HERE, "keys_enc_padmask.json"), "w"))
    enc.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes("keys_enc_padmask.json"), 930))
    enc.eval()
    tok = c["xq"].clone().requires_grad_(); kv = c["xkv"].clone().requires_grad_()
    y = enc(tok, c["pos"], kv, pad_mask=c["pad"])
    (y * c["R"]).sum().backward()
    out.update(enc_out=y.detach(), enc_dtok=tok.grad, enc_dkv=kv.grad)
    for k, v in enc.named_parameters():
        out["enc_g." + k] = v.grad
    save("padmask.npz", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "padmask":
        make_padmask()
    elif len(sys.argv) > 1 and sys.argv[1] == "ft":
        make_ft()
    elif len(sys.argv) > 1 and sys.argv[1] == "partseg":
        make_partseg()
    elif len(sys.argv) > 1 and sys.argv[1] == "augment":
        make_augment()
    elif len(sys.argv) > 1 and sys.argv[1] == "ckpt":
        make_ckpt()
    elif len(sys.argv) > 1 and sys.argv[1] == "ca2":
        make_ca2()
    elif len(sys.argv) > 1 and sys.argv[1] == "fullsize":        # python make_golden.py fullsize [c1 c3 c4]
        make_fullsize(tuple(sys.argv[2:]) or ("c1", "c3", "c4"))
    elif len(sys.argv) > 1 and sys.argv[1] == "partseg_full":
        make_partseg_full()
    elif len(sys.argv) > 2 and sys.argv[1] == "models":        # python make_golden.py models c3 c4
        main(only_models=tuple(sys.argv[2:]))
    else:
        main()
        make_ft()
        make_partseg()
        make_augment()
        make_ckpt()
        make_ca2()
        make_fullsize()
        make_partseg_full()
