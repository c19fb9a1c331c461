#!/usr/bin/env python3
"""FPS / index_points / kNN under a busy neighbour: the same call repeated, results compared bitwise with the first one.
usage: python tools/diag_fps_shared.py [runs=3000]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vipformer_amd.model.pointcloud import utils as U

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
torch.manual_seed(0)
if os.environ.get("SHIFT_VA"):          # move this process's allocations to other virtual addresses than a neighbour running the same framework uses
    _shift = torch.empty(int(os.environ["SHIFT_VA"]), dtype=torch.uint8, device="cuda")
pts = torch.randn(128, 1024, 3, device="cuda") * 0.3
print("pts at", hex(pts.data_ptr()))
start = torch.randint(0, 1024, (128,), device="cuda")
real = torch.randint
torch.randint = lambda *a, **k: start.clone()
def digest(t):
    return hash(t.cpu().numpy().tobytes())


seen = {"fps": collections.Counter(), "index_points": collections.Counter(), "centres": collections.Counter(), "groups": collections.Counter()}
keep = {}
idx_ref = U.farthest_point_sample(pts, 96)
for r in range(runs):
    idx = U.farthest_point_sample(pts, 96)
    ct = U.index_points(pts, idx_ref)
    nb, c = U.divide_patches(pts, 96, 32)
    torch.cuda.synchronize()
    for name, t in (("fps", idx), ("index_points", ct), ("centres", c), ("groups", nb)):
        d = digest(t)
        seen[name][d] += 1
        keep.setdefault((name, d), t.clone())
print(f"{runs} runs; the most frequent result vs all others (a deterministic kernel has one result):")
for name, cnt in seen.items():
    top, n = cnt.most_common(1)[0]
    print(f"   {name:13s} {n:5d} x the usual result, {runs - n:5d} x something else ({len(cnt)} distinct results)")
    if name == "fps" and len(cnt) > 1:
        ref = keep[(name, top)]
        for d, _ in cnt.most_common(4)[1:]:
            diff = keep[(name, d)] != ref
            rows = diff.any(1).nonzero().flatten().tolist()
            print(f"       clouds {rows[:6]}: first differing position {int(diff[rows[0]].nonzero()[0])}, {int(diff[rows[0]].sum())} of 96 positions differ in the first of them")
            b0, p0 = rows[0], int(diff[rows[0]].nonzero()[0])
            got = keep[(name, d)][b0]
            P = pts[b0].double()
            dmin = torch.full((1024,), 1e10, dtype=torch.float64, device="cuda")
            for g in range(p0):
                dmin = torch.minimum(dmin, ((P - P[ref[b0, g]]) ** 2).sum(1))
            order = torch.argsort(dmin, descending=True)
            rank = int((order == got[p0]).nonzero()[0]) + 1
            print(f"         cloud {b0} position {p0}: usual pick {int(ref[b0, p0])} (distance {float(dmin[ref[b0, p0]]):.6f}), this run picked {int(got[p0])} "
                  f"(distance {float(dmin[got[p0]]):.6f}, rank {rank} of 1024); picked before? {bool((ref[b0, :p0] == got[p0]).any())}")
