#!/usr/bin/env python3
"""bench.py -- pre-training pairs/s of the MI355X-native ViPFormer hot path.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A step = pretrain.py:173-211 semantics on one synthetic ShapeNetRender-shaped batch that is already
resident in HBM: zero_grad, point-cloud forward on cat(view1, view2) (FPS + kNN grouping + Group2Emb +
CA/SA encoder + head), image forward, two NT-Xent losses, backward, gradient all-reduce (RCCL) when
N > 1, fused AdamW.  Workload at every N: BASELINE.json configs[1] per GPU (E1CL6SL-H4D256-L96-MR2,
64 pairs of 2 x 1024-point clouds + one 224x224 image, patch 16) -> weak scaling, global batch 64 N.

Prints ONE JSON line (rank 0).  Besides the contract keys it carries
  roofline     : the dominant kernel (fused encoder-layer tail, HBM-bound) timed live with HIP events on the launch stream
  cpu_baseline : the oracle (CPU restatement of the reference path, kind "port") timed on this box's
                 host cores on a bounded sample (rank 0, N == 1 only)
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ARCH = dict(D=256, H=4, G=96, K=32, S=6, MR=2, N=1024, img=224, patch=16)     # configs[1] / c2
PER_GPU_PAIRS = 64
GFLOP_PER_PAIR = 17.4          # fwd+bwd, 2 FLOP/MAC, backward = 2x forward
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (guide: MI355X_MICROARCH.md)


def synth_batch(b, N, img, seed, device):
    """ShapeNetRender-shaped synthetic pairs: two augmented views of a unit-sphere cloud
    (independent noise clouds, each centred and scaled like PointcloudNormalize) + a ~N(0,1) image."""
    g = torch.Generator(device="cpu").manual_seed(seed)

    def cloud():
        p = torch.randn(b, N, 3, generator=g)
        p = p - p.mean(1, keepdim=True)
        return p / p.norm(dim=2).max(dim=1)[0].view(b, 1, 1)

    t1, t2 = cloud(), cloud()
    imgs = torch.randn(b, 3, img, img, generator=g)
    return t1.to(device), t2.to(device), imgs.to(device)


# Algorithmic HBM bytes of one row (token) through sa_layer_fwd_kernel: reads o (bf16 256) + residual base
# (f32 256) + pos (f32 256); writes x1, out (f32 256 each), n2, next n1 (bf16 256 each), u, h (bf16 512 each),
# next qkv (bf16 768), 4 LayerNorm statistics (f32).  Weights (1.3 MB per launch, L2 resident) are not counted.
SA_FWD_BYTES_PER_ROW = 256 * 2 + 256 * 4 + 256 * 4 + 2 * 256 * 4 + 2 * 256 * 2 + 2 * 512 * 2 + 768 * 2 + 16
SA_FWD_FLOP_PER_ROW = 2 * 256 * (256 + 512 + 512 + 768)
PEAK_HBM_GBS = 8000.0          # MI355X HBM3E (guide: MI355X_MICROARCH.md)


def time_dominant_kernel(device):
    """Roofline leg: the kernel with the largest share of the step (rocprofv3: profiles/r01_bench_c2_kernel_stats.csv),
    sa_layer_fwd_kernel -- the fused tail of one encoder layer (o_proj + dropout + residual, LayerNorm, MLP, next layer's
    LayerNorm + q/k/v projection) over the point-cloud branch's 2 * 64 * 96 = 12288 tokens -- launched back-to-back on
    torch's current stream (the stream the C ABI launches on) and timed with HIP events on that stream."""
    import torch.nn as nn
    from vipformer_amd import ops
    from vipformer_amd.model.pointcloud.partseg import SelfAttentionLayer
    B, Lq, D = 2 * PER_GPU_PAIRS, ARCH["G"], ARCH["D"]
    M = B * Lq
    layers = nn.ModuleList([SelfAttentionLayer(ARCH["H"], D, ARCH["MR"], 0.0, 0.1, 0.5) for _ in range(2)]).to(device)
    layers.train()
    blocks = [(l[0].module.attention, l[1].module, True, True) for l in layers]
    packed = ops._pack_blocks(blocks, layers[0], device)
    st = ops.rng.state(device)
    base = torch.randn(M, D, device=device)
    pos = torch.randn(M, D, device=device)
    o = torch.randn(M, D, device=device).to(torch.bfloat16)
    lse = torch.zeros(B * ARCH["H"] * Lq, device=device)
    att, mlp = layers[0][0].module.attention, layers[0][1].module
    nxt = (layers[1][0].module.norm, packed[1]["Wqkv"])

    def launch():
        ops._tail_fwd(att, mlp, layers[0][0], layers[0][1], packed[0], True, st, B, Lq, o, base, o, lse, nxt, pos, M, device)

    for _ in range(3):
        launch()
    iters = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = float(SA_FWD_BYTES_PER_ROW) * M
    ach = nbytes / (ms * 1e-3) / 1e9
    traffic = None
    try:        # HBM bytes per launch from the PMC passes of the same kernel (tools/pmc_hbm.sh), measured offline
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic_sa_layer_fwd.json")) as f:
            traffic = json.load(f)["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return dict(bound="hbm", kernel="sa_layer_fwd_kernel<2,2,32,false> (fused encoder-layer tail), 12288 tokens x 256 channels",
                achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(ach / PEAK_HBM_GBS, 4), traffic=traffic,
                us_per_launch=round(ms * 1e3, 2), bytes_per_launch=nbytes,
                mfma_tflops=round(SA_FWD_FLOP_PER_ROW * M / (ms * 1e-3) / 1e12, 1))


def cpu_baseline(pairs=4, timed_steps=2):
    """The oracle (fp32 torch-CPU restatement of the reference step, pinned against the reference by the
    golden fixtures) on this box's host cores: forward + backward + AdamW on `pairs` c1-shaped pairs."""
    from oracle import torch_oracle as O
    from tests import helpers as Hh
    a = ARCH
    arch = O.Arch(D=a["D"], H=a["H"], G=a["G"], K=a["K"], S=a["S"], MR=a["MR"], img=a["img"], patch=a["patch"])
    pc_sd = Hh.synth_state_dict(Hh.load_keyshapes("keys_pc_c1.json"), 100)
    im_sd = Hh.synth_state_dict(Hh.load_keyshapes("keys_img_c1.json"), 200)
    isparam = lambda k, v: v.dtype == torch.float32 and "running" not in k and "cross_attn_1." not in k
    pcp = {k: v.clone().requires_grad_() for k, v in pc_sd.items() if isparam(k, v)}
    imp = {k: v.clone().requires_grad_() for k, v in im_sd.items() if isparam(k, v)}
    s1, s2 = dict(pc_sd), dict(im_sd)
    s1.update(pcp); s2.update(imp)
    for s in (s1, s2):
        for k in list(s):
            if "cross_attn_1." in k:
                s[k] = s[k.replace("cross_attn_1.", "cross_attn_n.")]
    t1, t2, imgs = synth_batch(pairs, a["N"], a["img"], 0, "cpu")
    imgs = imgs.permute(0, 2, 3, 1)
    params = {**{"pc." + k: v for k, v in pcp.items()}, **{"img." + k: v for k, v in imp.items()}}
    state = {}
    times = []
    for it in range(1 + timed_steps):
        t0 = time.perf_counter()
        for v in params.values():
            v.grad = None
        start = torch.randint(0, a["N"], (2 * pairs,))
        loss, _, _ = O.pretrain_losses(s1, s2, t1, t2, imgs, start, arch, True, O.Masks("torch"), O.Masks("torch"), {}, {})
        loss.backward()
        with torch.no_grad():
            O.adamw_step({k: v for k, v in params.items()}, {k: v.grad for k, v in params.items()}, state, it + 1)
        if it > 0:
            times.append(time.perf_counter() - t0)
    sec = sum(times) / len(times)
    return dict(value=round(pairs / sec, 3), unit="pairs/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{timed_steps} timed steps (after 1 warm-up) of {pairs} pairs, E1CL6SL-H4D256-L96-MR2, 1024 pts + 224x224 "
                       f"img, fp32 torch-CPU oracle incl. FPS/kNN (C), fwd+bwd+AdamW; {sec:.2f} s/step")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run the image branch on the same stream as the point-cloud branch")
    ap.add_argument("--wgrad-async", action="store_true", help="grouped weight-gradient launches on a side stream")
    ap.add_argument("--pairs", type=int, default=PER_GPU_PAIRS, help="pairs per GPU (default = BASELINE configs[1])")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU with torch.distributed.run")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # "nccl" IS RCCL on ROCm

    from vipformer_amd import __version__, ops
    from vipformer_amd.train import Pretrainer, build_models

    torch.manual_seed(1)                                   # parser.py:17 default seed; same init on every rank
    ops.rng.seed(1234 + rank)
    pc, im = build_models(**ARCH, device=device)
    pc.train(); im.train()
    tr = Pretrainer(pc, im, world_size=world)
    tr.overlap = not args.no_overlap
    ops.WGRAD_GROUP_ASYNC[0] = args.wgrad_async
    tr.broadcast_parameters(0)
    torch.manual_seed(100 + rank)                          # FPS start indices differ per rank
    t1, t2, imgs = synth_batch(args.pairs, ARCH["N"], ARCH["img"], seed=rank, device=device)

    use_graph = not args.no_graph
    if use_graph:
        tr.capture(t1, t2, imgs, warmup=2)
        run = tr.replay
    else:
        run = lambda: tr.step(t1, t2, imgs)
    for _ in range(args.warmup):
        run()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = te.item()
    losses = [float(x) for x in tr.losses]
    finite = all(map(lambda v: v == v and abs(v) != float("inf"), losses))

    roof = time_dominant_kernel(device) if rank == 0 else None
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.pairs * world / (elapsed / args.steps)
        out = {
            "metric": "pretrain pairs/sec (1024 pts + 224^2 img, L96 H4D256)",
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "E1CL6SL-H4D256-L96-MR2, per-GPU batch %d pairs (2x1024-pt clouds + 224x224 img, patch 16), "
                                   "fwd+bwd+AdamW, NT-Xent IMC+CMC, dropout 0.1/0.5" % args.pairs,
                       "global_batch": args.pairs * world, "parallelism": f"dp{world}", "hip_graph": use_graph, "two_stream_overlap": tr.overlap,
                       "last_losses": losses, "losses_finite": finite,
                       "step_tflops_algorithmic": round(value * GFLOP_PER_PAIR / 1e3, 2),
                       "step_mfma_frac": round(value * GFLOP_PER_PAIR / 1e3 / (PEAK_BF16_TFLOPS * world), 4),
                       "version": __version__},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
