"""Which aten kernels does one training step still launch (eager, one stream)?  Prints op, shapes and the Python frame that issued it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vipformer_amd import ops
from vipformer_amd.train import Pretrainer, build_models
A = bench.ARCHS["c2"]
torch.manual_seed(1)
pc, im = build_models(**A, device="cuda")
pc.train(); im.train()
tr = Pretrainer(pc, im)
tr.overlap = False
t1, t2, imgs = bench.synth_batch(16, A["N"], A["img"], 0, "cuda")
for _ in range(2):
    tr.step(t1, t2, imgs)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    tr.step(t1, t2, imgs)
    torch.cuda.synchronize()
seen = {}
for e in prof.events():
    if not e.name.startswith("aten::"):
        continue
    if e.name in ("aten::empty", "aten::view", "aten::as_strided", "aten::empty_like", "aten::empty_strided", "aten::reshape", "aten::slice",
                  "aten::select", "aten::permute", "aten::detach", "aten::alias", "aten::_unsafe_view", "aten::contiguous", "aten::to",
                  "aten::_to_copy", "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::t", "aten::transpose", "aten::result_type",
                  "aten::zeros", "aten::clone", "aten::zeros_like", "aten::lift_fresh", "aten::item", "aten::_local_scalar_dense", "aten::is_nonzero"):
        continue
    frame = next((s for s in (e.stack or []) if "vipformer_amd" in s or "bench.py" in s), "?")
    key = (e.name, str(e.input_shapes)[:80], frame.strip()[:110])
    seen[key] = seen.get(key, 0) + 1
for (n, sh, fr), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"{c:3d} x {n:22s} {sh:80s} {fr}")
