import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import helpers as Hh
from vipformer_amd import ops
from vipformer_amd.train import Pretrainer, build_models

class forced_start:
    def __init__(self, s): self.s = s
    def __enter__(self):
        self.real = torch.randint; torch.randint = lambda *a, **k: self.s.clone()
    def __exit__(self, *a): torch.randint = self.real

a = Hh.ARCHS["c1"]; B = 4
t1 = Hh.synth_points(1, B, a["N"]).cuda(); t2 = Hh.synth_points(2, B, a["N"]).cuda()
imgs = Hh.synth_images(3, B, a["img"], a["img"]).permute(0, 3, 1, 2).contiguous().cuda()
start = Hh.synth_start(4, 2 * B, a["N"]).cuda()
res = []
for overlap in (False, True, False, True):
    ops.clear_managed_shadows(); ops.rng.seed(99); ops._site_counter[0] = 5000; torch.manual_seed(5)
    pc, im = build_models(D=a["D"], H=a["H"], G=a["G"], K=a["K"], S=a["S"], MR=a["MR"], N=a["N"], img=a["img"], patch=a["patch"])
    pc.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes("keys_pc_c1.json"), 100))
    im.load_state_dict(Hh.synth_state_dict(Hh.load_keyshapes("keys_img_c1.json"), 200))
    pc.train(); im.train()
    tr = Pretrainer(pc, im); tr.overlap = overlap
    with forced_start(start):
        losses = tr.forward_backward(t1, t2, imgs)
    torch.cuda.synchronize()
    names = [("pc." + k, p) for k, p in pc.named_parameters()] + [("img." + k, p) for k, p in im.named_parameters()]
    res.append((float(losses[0]), {k: p.grad.detach().clone() for k, p in names}))
def cos(x, y): return float((x.double().flatten() @ y.double().flatten()) / (x.double().norm() * y.double().norm() + 1e-30))
for i, j in ((0, 2), (1, 3), (0, 1)):
    worst = sorted((cos(res[i][1][k], res[j][1][k]), k) for k in res[i][1])[:6]
    print("variants", i, j, "loss", res[i][0], res[j][0], "worst:", [(round(c, 5), k) for c, k in worst])
