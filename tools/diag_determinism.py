import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import helpers as Hh
from vipformer_amd import ops
from vipformer_amd.model.pointcloud.utils import Group2Emb
from vipformer_amd.model.pointcloud import partseg as P
torch.manual_seed(0)
m = Group2Emb(256).cuda().train()
x = torch.randn(8, 96, 32, 3, device="cuda")
outs = []
for i in range(3):
    y = m(x); g = torch.ones_like(y)
    for p in m.parameters(): p.grad = None
    y.backward(g)
    outs.append((y.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
for i in (1, 2):
    d = (outs[i][0] - outs[0][0]).abs()
    print("g2e run", i, "out: differing elems", int((d > 0).sum()), "of", d.numel(), "max abs", float(d.max()), "rel", float(d.norm() / outs[0][0].norm()))
    for k in outs[0][1]:
        a, b = outs[0][1][k], outs[i][1][k]
        print("   grad", k, "rel diff %.2e" % float((a - b).norm() / (a.norm() + 1e-30)))
# one SA layer
torch.manual_seed(0)
layer = P.SelfAttentionLayer(4, 256, 2, 0.0, 0.1, 0.5).cuda().train()
xs = torch.randn(8, 96, 256, device="cuda", requires_grad=True)
res = []
for i in range(3):
    for p in layer.parameters(): p.grad = None
    xs.grad = None
    y = layer(xs); y.backward(torch.ones_like(y))
    res.append((y.detach().clone(), xs.grad.clone(), {k: p.grad.clone() for k, p in layer.named_parameters()}))
for i in (1, 2):
    print("sa run", i, "out equal", torch.equal(res[i][0], res[0][0]), "dx equal", torch.equal(res[i][1], res[0][1]),
          "max wgrad rel diff %.2e" % max(float((res[i][2][k] - res[0][2][k]).norm() / (res[0][2][k].norm() + 1e-30)) for k in res[0][2]))
