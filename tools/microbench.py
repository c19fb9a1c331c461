"""Per-kernel timings on the GPU box (HIP events on the launch stream).  Scratch tool."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import helpers as Hh
from tests.helpers import H16


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def preproc():
    from vipformer_amd.model.pointcloud import utils as U
    for (B, N, G, K) in [(128, 1024, 96, 32), (64, 1024, 128, 32), (32, 2048, 128, 32)]:
        pts = Hh.synth_points(1, B, N).cuda(); start = Hh.synth_start(1, B, N).cuda()
        t_fps = timeit(lambda: U._fps_from_start(pts, G, start))
        idx = U._fps_from_start(pts, G, start); ct = U.index_points(pts, idx)
        t_knn = timeit(lambda: U._knn_group(pts, ct, K, True, False, False, True))
        fps_bytes = B * (12 * N + 8 * G); knn_bytes = B * (12 * N + 12 * G + 12 * G * K + 12 * G)
        print(f"preproc B={B} N={N} G={G}: fps {t_fps:.1f} us ({t_fps/G*1e3:.0f} ns/iter, {fps_bytes/t_fps/1e3:.2f} GB/s)  "
              f"knn+group {t_knn:.1f} us ({knn_bytes/t_knn/1e3:.2f} GB/s)")




def attn():
    from vipformer_amd import _lib as L, ops
    st = ops.rng.state("cuda")
    for (B, H, Lq, Lkv, p, tag) in [(64, 4, 196, 196, 0.1, "img"), (128, 4, 96, 96, 0.1, "pc-SA"), (128, 4, 96, 1024, 0.1, "pc-CA")]:
        D = 64 * H
        q = torch.randn(B * Lq, D, device="cuda").to(H16); k = torch.randn(B * Lkv, D, device="cuda").to(H16)
        v = torch.randn(B * Lkv, D, device="cuda").to(H16); do = torch.randn(B * Lq, D, device="cuda").to(H16)
        o = torch.empty_like(q); lse = torch.empty(B * H * Lq, device="cuda")
        dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
        f = lambda: L.call("vpf_attention_fwd", q, D, k, D, v, D, B, H, Lq, Lkv, 64, 0.125, p, st, 7, o, D, lse)
        b = lambda: L.call("vpf_attention_bwd", q, D, k, D, v, D, o, D, do, D, lse, B, H, Lq, Lkv, 64, 0.125, p, st, 7, dq, D, dk, D, dv, D, torch.empty(B * H * Lq, dtype=torch.float32, device="cuda"))
        tf, tb = timeit(f, 20, 3), timeit(b, 20, 3)
        fl = 4.0 * B * H * Lq * Lkv * 64
        print(f"attn {tag} B={B} H={H} Lq={Lq} Lkv={Lkv}: fwd {tf:.1f} us ({fl/tf/1e6:.1f} TF/s)  bwd {tb:.1f} us ({2.5*fl/tb/1e6:.1f} TF/s)")


def gemm():
    from vipformer_amd import ops
    for (M, N, K, tag) in [(393216, 256, 256, "g2e conv3"), (393216, 128, 64, "g2e conv2"), (131072, 512, 256, "CA kv proj"),
                           (12288, 768, 256, "SA qkv"), (12288, 512, 256, "fc1"), (12288, 256, 512, "fc2"), (12544, 256, 768, "patch")]:
        A = torch.randn(M, K, device="cuda").to(H16); W = (torch.randn(N, K, device="cuda") * 0.05).to(H16)
        dY = torch.randn(M, N, device="cuda").to(H16); dW = torch.zeros(N, K, device="cuda")
        bias = torch.zeros(N, device="cuda")
        t1 = timeit(lambda: ops.linear_fwd(A, W, N, K, bias), 20, 3)
        t2 = timeit(lambda: ops.linear_dgrad(dY, W, N, K), 20, 3)
        t3 = timeit(lambda: ops.linear_wgrad(dY, A, N, K, dW), 20, 3)
        fl = 2.0 * M * N * K
        print(f"gemm {tag} M={M} N={N} K={K}: fwd {t1:.1f} us ({fl/t1/1e6:.0f} TF/s) dgrad {t2:.1f} us ({fl/t2/1e6:.0f} TF/s) wgrad {t3:.1f} us ({fl/t3/1e6:.0f} TF/s)")


def wgrad():
    from vipformer_amd import ops
    for (M, N, K, tag) in [(12288, 256, 512, "fc2"), (12288, 512, 256, "fc1"), (12288, 256, 256, "proj"), (12288, 768, 256, "qkv"),
                           (131072, 512, 256, "CA kv")]:
        A = torch.randn(M, K, device="cuda").to(H16)
        dY = torch.randn(M, N, device="cuda").to(H16); dW = torch.zeros(N, K, device="cuda")
        t3 = timeit(lambda: ops.linear_wgrad(dY, A, N, K, dW), 300, 10)
        fl = 2.0 * M * N * K
        print(f"wgrad {tag} M={M} N={N} K={K}: {t3:.1f} us ({fl/t3/1e6:.0f} TF/s)")


def sa():
    """fused self-attention layer forward: time per layer + phase cycles, pc (L=96) and img (L=196) shapes"""
    from vipformer_amd import ops
    from vipformer_amd.model.pointcloud.partseg import SelfAttentionLayer
    import torch.nn as nn
    for (B, Lq, tag) in [(128, 96, "pc"), (64, 196, "img")]:
        layers = nn.ModuleList([SelfAttentionLayer(4, 256, 2, 0.0, 0.1, 0.5) for _ in range(6)]).cuda()
        layers.train()
        x = torch.randn(B, Lq, 256, device="cuda"); pos = torch.randn(B if tag == "pc" else 1, Lq, 256, device="cuda")
        params = [p for l in layers for p in l.parameters()]
        dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
        for fused, split in ((False, None), (True, False), (True, True)):
            ops.cfg.sa_fused = fused
            ops.cfg.sa_split_attn = split
            def run():
                with torch.no_grad():
                    if fused:
                        return ops.SAStackFn.apply(x, pos, layers, True, *params)
                    y = x
                    for l in layers: y = l(y, pos=pos)
                    return y
            t = timeit(run, 20, 3)
            print(f"sa stack fwd {tag} B={B} L={Lq} fused={fused} split_attn={split}: {t:.1f} us / 6 layers = {t/6:.1f} us per layer")
        ops.cfg.sa_debug = dbg
        run(); torch.cuda.synchronize()
        ops.cfg.sa_debug = None
        names = ["attention", "o_proj mfma", "drop+res epi", "LN2 maths", "n2 stores", "barrier", "chunk0+fc1(1)", "bias+u+gelu", "barrier", "h stores", "barrier", "fc2(1)", "MLP rest", "final epi", "next LN1", "next qkv"]
        print("   phase cycles (wg 0, layer 4): " + "  ".join(f"{n} {int(c)}" for n, c in zip(names, dbg.tolist())))
    ops.cfg.sa_fused = True
    ops.cfg.sa_split_attn = None


def satail():
    """the fused encoder-layer tail exactly as bench.py's kernel leg launches it (pc shape: 12288 tokens)"""
    import torch.nn as nn
    from vipformer_amd import ops
    from vipformer_amd.model.pointcloud.partseg import SelfAttentionLayer
    B, G, D, H = 128, 96, 256, 4
    M = B * G
    layers = nn.ModuleList([SelfAttentionLayer(H, D, 2, 0.0, 0.1, 0.5) for _ in range(2)]).cuda()
    layers.train()
    blocks = [(l[0].module.attention, l[1].module, True, True) for l in layers]
    packed = ops._pack_blocks(blocks, layers[0], "cuda")
    st = ops.rng.state("cuda")
    base = torch.randn(M, D, device="cuda"); pos = torch.randn(M, D, device="cuda")
    o = torch.randn(M, D, device="cuda").to(H16)
    lse = torch.zeros(B * H * G, device="cuda")
    att, mlp = layers[0][0].module.attention, layers[0][1].module
    nxt = (layers[1][0].module.norm, packed[1]["Wqkv"])
    t = timeit(lambda: ops._tail_fwd(att, mlp, layers[0][0], layers[0][1], packed[0], True, st, B, G, o, base, o, lse, nxt, pos, M, "cuda"), 20, 3)
    print(f"sa_layer_fwd tail (12288 tokens): {t:.1f} us")
    dbg = torch.zeros(32 + 4 * 2048, dtype=torch.int64, device="cuda")
    ops.cfg.sa_debug = dbg
    ops._tail_fwd(att, mlp, layers[0][0], layers[0][1], packed[0], True, st, B, G, o, base, o, lse, nxt, pos, M, "cuda")
    torch.cuda.synchronize()
    ops.cfg.sa_debug = None
    names = ["loads+barrier", "o_proj", "epi1+x1 out", "LN2+tile+barrier", "n2 out", "fc1(0)", "u+gelu pass(0)", "barrier", "fc2(0)",
             "chunk1 fc1..gelu + x1,pos in", "fc2(1)", "epi2+out", "LN1n+tile+barrier", "n1 out", "qkv"]
    c = dbg.tolist()
    print("   phase cycles (wg 0): " + "  ".join(f"{n} {int(v)}" for n, v in zip(names, c)) + f"  total {sum(c[:15])}")
    rec = dbg[32:].view(-1, 4).cpu()
    last = rec[1024:]
    rec = rec[:1024]
    keep = rec[:, 1] > 0
    if bool((last[:, 1] > 0).any()):
        lag = (last[:len(rec)][keep][:, 1] - rec[keep][:, 1]).double() / 100.0
        span = (int(last[:, 1].max()) - int(rec[keep][:, 0].min())) / 100.0
        print(f"   last group's end minus first group's end: median {lag.median():.1f} us, max {lag.max():.1f} us; first start -> last group's last end {span:.1f} us")
    rec = rec[keep]
    if len(rec):
        import collections
        t0 = int(rec[:, 0].min())
        per_cu = collections.Counter((int(r[2]), int(r[3]) & 0xff00) for r in rec)      # (XCC, HW_ID without the wave-slot bits)
        dur = (rec[:, 1] - rec[:, 0]).double() / 100.0
        print(f"   {len(rec)} workgroups: duration us min {dur.min():.1f} median {dur.median():.1f} max {dur.max():.1f}; first start -> last end "
              f"{(int(rec[:, 1].max()) - t0) / 100.0:.1f} us; start spread {(int(rec[:, 0].max()) - t0) / 100.0:.1f} us; "
              f"workgroups per (XCC, CU): {sorted(collections.Counter(per_cu.values()).items())}")
        if os.environ.get("WG_DUMP"):
            order = sorted(range(len(rec)), key=lambda i: int(rec[i, 0]))
            for i in order[:: max(1, len(order) // 48)]:
                r = rec[i]
                print(f"      block {i:4d} start {(int(r[0]) - t0) / 100.0:6.2f} us  dur {(int(r[1]) - int(r[0])) / 100.0:6.2f}  xcc {int(r[2])} hw_id {int(r[3]) & 0xffff:#06x}")
        shared = [i for i, r in enumerate(rec) if per_cu[(int(r[2]), int(r[3]) & 0xff00)] > 1]
        alone = [i for i, r in enumerate(rec) if per_cu[(int(r[2]), int(r[3]) & 0xff00)] == 1]
        if shared and alone:
            print(f"   alone on a CU: median {dur[alone].median():.1f} us ({len(alone)}); sharing a CU: median {dur[shared].median():.1f} us ({len(shared)})")


def satail3():
    """the forward tail at 96 / 192 / 256 row blocks of 64 tokens (half a chip, the step's grid, one per CU): is a launch's time set by
    what ONE CU can do (flat in the grid size) or by chip-level traffic (proportional)?   SATAIL_B=48,96,128 overrides the batch list"""
    import torch.nn as nn
    from vipformer_amd import ops
    from vipformer_amd.model.pointcloud.partseg import SelfAttentionLayer
    G, D, H = 96, 256, 4
    for B in [int(v) for v in os.environ.get("SATAIL_B", "64,128,170").split(",")]:
        M = B * G
        layers = nn.ModuleList([SelfAttentionLayer(H, D, 2, 0.0, 0.1, 0.5) for _ in range(2)]).cuda()
        layers.train()
        blocks = [(l[0].module.attention, l[1].module, True, True) for l in layers]
        packed = ops._pack_blocks(blocks, layers[0], "cuda")
        st = ops.rng.state("cuda")
        base = torch.randn(M, D, device="cuda"); pos = torch.randn(M, D, device="cuda")
        o = torch.randn(M, D, device="cuda").to(H16)
        lse = torch.zeros(B * H * G, device="cuda")
        att, mlp = layers[0][0].module.attention, layers[0][1].module
        nxt = (layers[1][0].module.norm, packed[1]["Wqkv"])
        t = timeit(lambda: ops._tail_fwd(att, mlp, layers[0][0], layers[0][1], packed[0], True, st, B, G, o, base, o, lse, nxt, pos, M, "cuda"), 30, 5)
        print(f"sa_layer_fwd tail, {M} tokens = {M // 64} row blocks: {t:.1f} us")


def satail2():
    """Two encoder-layer tails side by side, as the step runs them: the point-cloud shape (128 x 96 tokens) on one stream and the image
    shape (64 x 196) on another, six launches each, captured into ONE hipGraph and replayed -- with the round-2 kernels (one
    workgroup per CU by LDS) and with the round-3 row-block kernels in every geometry (VPF_SA_WG2 / VPF_SA_RB as debug knobs).
    Answers whether workgroups of two concurrent launches sharing CUs buy what the stand-alone numbers promise."""
    import torch.nn as nn
    from vipformer_amd import _lib as L
    from vipformer_amd import ops
    from vipformer_amd.model.pointcloud.partseg import SelfAttentionLayer
    D, H = 256, 4
    shapes = [(128, 96), (64, 196)]
    sets = []
    for B, G in shapes:
        M = B * G
        layers = nn.ModuleList([SelfAttentionLayer(H, D, 2, 0.0, 0.1, 0.5) for _ in range(2)]).cuda()
        layers.train()
        blocks = [(l[0].module.attention, l[1].module, True, True) for l in layers]
        packed = ops._pack_blocks(blocks, layers[0], "cuda")
        base = torch.randn(M, D, device="cuda"); pos = torch.randn(G, D, device="cuda")
        o = torch.randn(M, D, device="cuda").to(H16)
        lse = torch.zeros(B * H * G, device="cuda")
        sets.append((B, G, M, layers, packed, base, pos, o, lse))
    st = ops.rng.state("cuda")

    def tail(s):
        B, G, M, layers, packed, base, pos, o, lse = s
        att, mlp = layers[0][0].module.attention, layers[0][1].module
        nxt = (layers[1][0].module.norm, packed[1]["Wqkv"])
        return ops._tail_fwd(att, mlp, layers[0][0], layers[0][1], packed[0], True, st, B, G, o, base, o, lse, nxt, pos, G, "cuda")

    def run(which, n=6):
        keep = []
        main = torch.cuda.current_stream()
        side = [torch.cuda.Stream() for _ in which]
        for sd in side:
            sd.wait_stream(main)
        for sd, w in zip(side, which):
            with torch.cuda.stream(sd):
                for _ in range(n):
                    keep.append(tail(sets[w]))
        for sd in side:
            main.wait_stream(sd)
        return keep

    def timed(which, label):
        run(which); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        cs = torch.cuda.Stream()
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            with torch.cuda.graph(g, stream=cs):
                keep = run(which)
        torch.cuda.current_stream().wait_stream(cs)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"   {label:34s} {us:8.1f} us per graph = {us / 6:6.1f} us per layer slot")
        del keep
        return us

    for name, wg2, rb in (("round-2 kernels (1 WG/CU)", 0, 0), ("sa_rows <2,1> 64 tok, 76 KB", 1, 2), ("sa_rows <1,1> 32 tok cut to 2/CU", 1, 0),
                          ("sa_rows <1,1> 32 tok whole", 1, 1), ("sa_rows <1,2> 64 tok 16 waves", 1, 12),
                          ("sa_rows <1,2> DECOUPLED groups", 1, 13), ("sa_rows <1,2> DECOUPLED, stagger 20", 1, 13 + 20 * 256),
                          ("sa_rows <2,1> cut to 1 per CU (48 / 49 tokens)", 1, 2 + 65536), ("sa_rows <1,1> cut to 3 per CU (16 / 17 tokens)", 1, 1 + 3 * 65536)):
        L.debug_set("sa_stagger", (rb >> 8) & 255); L.debug_set("sa_tpw", -(rb >> 16)); rb &= 255
        if os.environ.get("SATAIL2_ONLY") and str(rb) not in os.environ["SATAIL2_ONLY"].split(","):
            continue
        L.debug_set("sa_wg2", wg2); L.debug_set("sa_rb", rb)
        print(name)
        a = timed([0], "pc alone (6 launches)")
        b = timed([1], "img alone (6 launches)")
        c = timed([0, 1], "pc || img (two streams)")
        print(f"   -> side by side / (pc + img alone) = {c / (a + b):.3f}")
    L.debug_set("sa_wg2", 0); L.debug_set("sa_rb", 0); L.debug_set("sa_tpw", 0); L.debug_set("sa_stagger", 0)


def wgroup():
    """the grouped weight-gradient launch of one encoder layer (4 problems, M = 12288 tokens)"""
    from vipformer_amd import ops
    M = 12288
    shapes = [(256, 512), (512, 256), (256, 256), (768, 256)]
    ts = []
    for (N, K) in shapes:
        ts.append((torch.randn(M, N, device="cuda").to(H16), torch.randn(M, K, device="cuda").to(H16), N, K,
                   torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")))
    def run():
        wg = ops.WgradBatch()
        for dy, x, N, K, dW, db in ts: wg.add(dy, x, N, K, dW, None if os.environ.get("NOBIAS") else db)
        wg.flush()
    t = timeit(run, 100, 5)
    fl = sum(2.0 * M * N * K for _, _, N, K, _, _ in ts)
    print(f"wgrad group (4 problems, M={M}): {t:.1f} us ({fl/t/1e6:.0f} TF/s)")


def wstack():
    """the grouped weight-gradient launch the STEP makes: the 28 weight gradients of the point-cloud encoder stack (bench.py's leg);
    read its device-side duration from a kernel trace (tools/kprof.sh): the eager launch is host-bound (28 job descriptors)"""
    from vipformer_amd import ops
    M, D, Hd, S = 12288, 256, 512, 6
    shapes = [(D, Hd), (Hd, D), (D, D), (3 * D, D)]
    stack = [(D, Hd), (Hd, D), (D, D), (D, D)] + [sh for _ in range(S) for sh in shapes]
    g = torch.Generator().manual_seed(7)
    jobs = [(torch.randn(M, N, generator=g).cuda().to(H16), torch.randn(M, K, generator=g).cuda().to(H16), N, K,
             torch.zeros(N, K, device="cuda"), (torch.zeros(N, device="cuda") if i % 4 != 3 else None)) for i, (N, K) in enumerate(stack)]
    def run():
        wg = ops.WgradBatch(cap=ops.WgradBatch.CAP)
        for dy, x, N, K, dW, db in jobs: wg.add(dy, x, N, K, dW, db)
        wg.flush()
    t = timeit(run, 20, 3)
    fl = sum(2.0 * M * N * K for N, K in stack)
    print(f"wgrad stack ({len(stack)} problems, M={M}): {t:.1f} us eager ({fl/t/1e6:.0f} TF/s)")


def g2e():
    """Group2Emb forward/backward at the benchmark size + per-phase cycle stamps of the fused backward."""
    from vipformer_amd import ops
    from vipformer_amd.model.pointcloud.utils import Group2Emb
    torch.manual_seed(0)
    m = Group2Emb(256).cuda().train()
    x = torch.randn(128, 96, 32, 3, device="cuda")
    y = m(x); g = torch.randn_like(y)
    def step():
        y = m(x); y.backward(g)
    for _ in range(3): step()
    torch.cuda.synchronize()
    print("g2e fwd+bwd %.1f us" % timeit(step, 10, 2))
    dbg = torch.zeros(256 * 2 * 6, dtype=torch.int64, device="cuda")
    ops.G2E_DEBUG["dbg"] = dbg
    step(); torch.cuda.synchronize()
    d = dbg.view(256, 2, 6).double().mean(0)
    names = ["stage+barrier", "mfma da3", "bn epilogue+barrier", "dh3 store", "mfma dh2 + stage + barrier", "dh2 store + barrier"]
    for ps in range(2):
        tot = d[ps].sum().item()
        print("pass", ps, "cycles/WG %.0f:" % tot, ", ".join(f"{n} {v/tot*100:.0f}%" for n, v in zip(names, d[ps].tolist())))
    ops.G2E_DEBUG.clear()


def gemmk():
    """Sensitivity of the small GEMM to K (fixed M=12288, N=256) and to the output type: separates the fixed
    (launch + prologue + epilogue) cost from the per-k-step cost."""
    from vipformer_amd import ops
    M, N = 12288, 256
    for K in (32, 64, 128, 256, 512, 1024, 2048):
        A = torch.randn(M, K, device="cuda").to(H16); W = (torch.randn(N, K, device="cuda") * 0.05).to(H16)
        y16 = torch.empty(M, N, dtype=H16, device="cuda"); y32 = torch.empty(M, N, dtype=torch.float32, device="cuda")
        t1 = timeit(lambda: ops.gemm(A, 0, K, W, 0, K, M, N, K, y16, N, c_f32=False), 30, 5)
        t2 = timeit(lambda: ops.gemm(A, 0, K, W, 0, K, M, N, K, y32, N, c_f32=True), 30, 5)
        print(f"gemmk K={K}: h16-out {t1:.1f} us  f32-out {t2:.1f} us")
    e = torch.empty(8, device="cuda")
    print("empty launch (cast of 8 elems): %.1f us" % timeit(lambda: ops.to_h16(e), 50, 5))


def gemmtn():
    """the single weight-gradient GEMMs of the step (dW[N,K] += dY^T X over M rows, split over M, fp32 atomics) per tile configuration:
    VPF_WGRAD_CFG 0 = 64x64, 1 = 128x64, 2 = 128x128; and per split (0 = the launcher's own choice).  The cfg sweep runs with the LDS-DMA
    kernel OFF (wgroup_dma = 0: at splitk 0 a conforming shape would otherwise take gemm_wgrad_dma_kernel whatever cfg says and the
    three columns would time the same kernel -- ADVICE r05); the DMA kernel gets a column of its own."""
    from vipformer_amd import _lib as L
    from vipformer_amd import ops
    shapes = [("g2e dW3[:,128:] = dh3^T h2", 393216, 256, 128), ("g2e conv2 dW = dh2^T a1", 393216, 128, 64),
              ("patch embedding dW = dy^T patches", 12544, 256, 768), ("g2e dW3[:,:128] = dgb^T gmax", 12288, 256, 128),
              ("adapter l3 dW = dy^T a1", 131072, 256, 64), ("kv dW = dkv^T nk", 131072, 512, 256)]
    for name, M, N, K in shapes:
        dy = torch.randn(M, N, device="cuda").to(H16); x = torch.randn(M, K, device="cuda").to(H16)
        dW = torch.zeros(N, K, device="cuda")
        line = f"{name:38s} M={M:6d} N={N:3d} K={K:3d}:"
        dma_before = L.debug_get("wgroup_dma")
        t = timeit(lambda: ops.gemm(dy, 1, N, x, 1, K, N, K, M, dW, K, c_f32=True, mode=ops.EPI_ATOMIC, splitk=0), 20, 3)
        line += f"  shipped (DMA kernel where the shape conforms) {t:6.1f} |"
        L.debug_set("wgroup_dma", 0)
        for cfg in (0, 1, 2):
            L.debug_set("wgrad_cfg", cfg)
            for sk in (0,) + ((128, 512) if M > 100000 else (16, 32)):
                t = timeit(lambda: ops.gemm(dy, 1, N, x, 1, K, N, K, M, dW, K, c_f32=True, mode=ops.EPI_ATOMIC, splitk=sk), 20, 3)
                line += f"  cfg{cfg}/sk{sk} {t:6.1f}"
        L.debug_set("wgrad_cfg", 0)
        L.debug_set("wgroup_dma", dma_before)
        gb = M * (N + K) * 2 / 1e9
        print(line + f"   us   ({gb * 1e3:.0f} MB: {gb / 6.3e3 * 1e6:.1f} us at 6.3 TB/s)")


if __name__ == "__main__":
    which = sys.argv[1:] or ["preproc"]
    for w in which:
        globals()[w]()


def _one_gemm(kind):
    """neighbour loads for tools/diag_fps_shared.py: one GEMM flavour in a loop"""
    from vipformer_amd import ops
    M, N, K = 12288, 512, 256
    A = torch.randn(M, K, device="cuda").to(H16); W = (torch.randn(N, K, device="cuda") * 0.05).to(H16)
    dY = torch.randn(M, N, device="cuda").to(H16); dW = torch.zeros(N, K, device="cuda")
    for _ in range(200):
        if kind == "fwd": ops.linear_fwd(A, W, N, K, None)
        elif kind == "dgrad": ops.linear_dgrad(dY, W, N, K)
        else: ops.linear_wgrad(dY, A, N, K, dW)
    torch.cuda.synchronize()


def gemm_fwd(): _one_gemm("fwd")
def gemm_dgrad(): _one_gemm("dgrad")
def gemm_wgrad(): _one_gemm("wgrad")
