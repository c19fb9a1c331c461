"""Stage-by-stage comparison of the part-segmentation head (HIP) against the h16-emulating oracle (diagnostic, GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tests import helpers as Hh
from tests.test_modules_gpu import forced_start, rel
from tests.test_partseg_gpu import _build, _inputs
from oracle import torch_oracle as O
from vipformer_amd import ops, ops_seg as S

name = sys.argv[1] if len(sys.argv) > 1 else "tinyseg"
train = (sys.argv[2] if len(sys.argv) > 2 else "train") == "train"
a, B, pts, start, cls, target = _inputs(name)
lidx = Hh.PARTSEG_LAYERS[name]
m = _build(name)
m.train(train)
m.dp1.p = 0.0
sd = Hh.synth_state_dict(Hh.load_keyshapes(f"keys_partseg_{name}.json"), 100)
arch = O.Arch(D=a["D"], H=a["H"], G=a["G"], K=a["K"], S=a["S"], MR=a["MR"], img=a["img"], patch=a["patch"], atten_drop=0.0, mlp_drop=0.0)
with torch.no_grad(), forced_start(start.cuda()):
    feats, center = m._encode(pts.cuda(), lidx)
    nl = len(feats)
    xcat = S.LnTapsFn.apply(m.norm, nl, *feats, *m.norm.parameters())
    pooled = ops.PoolFn.apply(xcat)
    lf = S.LabelBranchFn.apply(cls.cuda(), m.label_conv, train, *m.label_conv.parameters())
    f0 = S.FeaturePropFn.apply(pts.cuda(), center, pts.cuda(), xcat, m.propagation, train, *m.propagation.parameters())
    gvec = torch.cat([pooled, lf], 1)
    head = [m.conv1, m.bn1, m.conv2, m.bn2, m.conv3]
    y = S.SegConvFn.apply(f0, gvec, m, train, *[p for mm in head for p in mm.parameters()])
for emu in (False, True):
    ctxm = O.emulate_bf16() if emu else torch.no_grad()
    with torch.no_grad(), ctxm:
        masks = O.Masks("off")
        kv = O.adapter(sd, "input_adapter.", pts)
        fidx = O.fps_indices(pts, start, arch.G)
        nb, ct, _ = O.divide_patches(pts, fidx, arch.K, True)
        tok = O.group2emb(sd, "group2emb.", nb, train, {})
        pos = O.pos_mlp(sd, "position_emb.", ct)
        _, fr = O.encoder(sd, "encoder.", tok, pos, kv, arch, masks, taps=tuple(lidx))
        D = arch.D
        x = torch.cat([F.layer_norm(f, (D,), sd["norm.weight"], sd["norm.bias"], 1e-5) for f in fr], dim=2)
        x_max, x_avg = x.max(dim=1)[0], x.mean(dim=1)
        lab = F.conv1d(O.Q(cls.view(B, 16, 1)), O.Q(sd["label_conv.0.weight"]))
        lab = O.Q(F.leaky_relu(O._bn(sd, "label_conv.1.", lab, train, {}), 0.2)).view(B, 64)
        f0r = O.feature_propagation(sd, "propagation.", pts, ct, pts, x, train, {})
        yr = O.partseg_forward(sd, pts, start, cls, arch, lidx, train, masks, {}, head_mask=torch.full((B * a["N"], 512), 0.5))
    tag = "emulated" if emu else "fp32"
    for i in range(nl):
        print(f"[{tag}] tap {lidx[i]} rel {rel(feats[i], fr[i]):.3e}")
    print(f"[{tag}] centers equal {torch.equal(center.cpu(), ct)}  xcat rel {rel(xcat, x):.3e}  pooled rel {rel(pooled, torch.cat([x_max, x_avg], 1)):.3e}  "
          f"label rel {rel(lf, lab):.3e}  f_level_0 rel {rel(f0, f0r):.3e}  logits rel {rel(y, yr):.3e}")
