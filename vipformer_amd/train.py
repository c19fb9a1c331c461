"""The --mp pre-training step of the reference (pretrain.py:173-211) on MI355X.

    zero_grad -> pc forward on cat(view1, view2) -> NT-Xent(view1, view2)  (IMC)
              -> img forward -> NT-Xent(mean(view1, view2), img)          (CMC)
              -> backward -> gradient all-reduce (data parallel) -> AdamW

What is specific to this implementation:
  * all parameters of both models live in ONE flat fp32 buffer (q/k/v weights adjacent so the
    fused [3D,D] projection needs no copy); gradients in one flat fp32 buffer that the wgrad
    kernels accumulate into directly; Adam moments flat; a flat h16 shadow of the weights is
    rewritten by the fused AdamW kernel (vpf_adamw_step) and is what the MFMA kernels read.
  * data parallelism = one process per GPU; the only collective is ONE all-reduce of the flat
    gradient buffer over RCCL/xGMI (the reference's two DDP reducers with 25 MB buckets become a
    single 33 MB message); NT-Xent negatives and BatchNorm statistics stay rank-local exactly
    like the reference (lightly 1.1.21 has no gather; no SyncBatchNorm).  The mean over ranks is
    folded into the AdamW kernel's gradient scale.
  * the MFMA operands are fp16 like the reference's autocast (pretrain.py:154,176), so the backward pass runs under
    GradScaler's loss scale (pretrain.py:154,209-211) -- kept ON THE DEVICE so that a captured hipGraph follows it: the
    loss gradient is seeded with the scale, vpf_grad_check looks for inf / NaN in the flat gradient, the AdamW kernel divides
    the scale out or skips the step, and the kernel behind it halves / doubles the scale (torch.cuda.amp.GradScaler's defaults:
    65536, x2 every 2000 good steps, x0.5 on overflow).
  * the whole step (forward, backward, AdamW, dropout-state advance) can be captured into one
    hipGraph (``capture=True``): a few hundred short kernels per step are launch-bound from Python.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import os

import torch
import torch.distributed as dist

from . import _lib as L
from . import ops


class FlatParams:
    """Flat fp32 parameters / gradients / Adam moments + h16 shadow for a list of modules."""

    def __init__(self, modules: Sequence[torch.nn.Module]):
        seen, params = set(), []
        for m in modules:
            for p in m.parameters():
                if id(p) not in seen:
                    seen.add(id(p))
                    params.append(p)
        self.params: List[torch.nn.Parameter] = params
        dev = params[0].device
        offs, n = [], 0
        for p in params:
            n = (n + 7) // 8 * 8            # 16-byte aligned h16 shadow / 32-byte aligned fp32
            offs.append(n)
            n += p.numel()
        n = (n + 7) // 8 * 8
        self.numel = n
        self.p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.s = torch.empty(n, dtype=L.H16, device=dev)
        self.managed = ops.ManagedFlat(self.p, self.s, self.g)      # parameters point at it weakly: no process-wide registry
        for p, o in zip(params, offs):
            self.p[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.p[o:o + p.numel()].view_as(p.data)
            p.grad = self.g[o:o + p.numel()].view_as(p.data)
            self.managed.adopt(p, o)
        self.offsets = offs
        self.refresh_shadow()

    def refresh_shadow(self) -> None:
        """Re-cast the whole h16 shadow from the fp32 values.  ops.shadow() does this per parameter when ``p._version`` moved
        (load_state_dict, an external optimizer); raw writes through ``p.data`` / ``flat.p`` do not bump it: call this then."""
        L.call("vpf_cast_f32_h16", self.p, self.s, self.numel)
        for p in self.params:
            p._vpf_ver = (p._version, ops._OPT_EPOCH[0])

    def attach_grads(self) -> None:
        """(Re-)install the flat views as ``p.grad`` wherever they are missing: after optimizer.zero_grad(set_to_none=True)
        (pretrain.py:174) a parameter that receives its gradient through autograd (the image model's ``position_emb``) would otherwise
        get a fresh tensor AdamW never sees."""
        base = self.g.data_ptr()
        for p, o in zip(self.params, self.offsets):
            g = p.grad
            if g is None or g.data_ptr() != base + 4 * o:
                p.grad = self.g[o:o + p.numel()].view_as(p.data)


def sync_module_buffers(modules, src: int = 0, group=None) -> int:
    """What DistributedDataParallel(broadcast_buffers=True) does in front of EVERY forward pass (pretrain.py:104-105 wrap the models;
    :186,199 training forwards, :243,266 the eval-mode probe forwards): rank `src`'s buffers -- the BatchNorm running statistics and
    step counters -- everywhere.  Works on CPU tensors over gloo as on HBM over RCCL.  Returns the number of buffers broadcast."""
    n = 0
    for m in modules:
        for b in m.buffers():
            dist.broadcast(b, src, group=group)
            n += 1
    return n


class GradientExchange:
    """The data-parallel exchange step (what the two DistributedDataParallel reducers of pretrain.py:104-105 do during
    ``backward``): SUM all-reduce of the flat gradient over the ranks, as independent REGIONS that are launched asynchronously on a
    communication stream the moment their producer has finished -- the image branch's gradients travel while the point-cloud
    branch's backward is still running, and AdamW of a region starts when that region has arrived.  The mean (1 / world) is folded
    into the AdamW kernel's gradient scale.  No kernels of this library are involved: the class works on CPU tensors over gloo
    (tests/test_host_cpu.py) exactly as on HBM over RCCL / xGMI."""

    def __init__(self, flat_g: torch.Tensor, regions, world: int, group=None, wire_bf16: bool = False, always: bool = False):
        self.g, self.world, self.group, self.wire_bf16 = flat_g, world, group, wire_bf16
        self.always = always          # run the collectives even in a one-rank group (tests: the RCCL code path on a single GPU)
        self.regions = [(str(n), int(a), int(b)) for n, a, b in regions]
        covered = sorted((a, b) for _, a, b in self.regions)
        if covered[0][0] != 0 or covered[-1][1] != flat_g.numel() or any(x[1] != y[0] for x, y in zip(covered[:-1], covered[1:])):
            raise ValueError("regions must tile the flat gradient buffer exactly")
        self._cuda = flat_g.is_cuda
        self._comm = torch.cuda.Stream(device=flat_g.device) if self._cuda else None
        self._pending = {}
        self.timing = False           # bench.py: device events around every region's exchange (comm_ms)
        self._events = []

    def start(self, name: str) -> None:
        """Launch the all-reduce of region ``name``; on a GPU it is ordered after everything queued so far on the CURRENT stream
        (the stream that produced the region) and runs on the communication stream."""
        if self.world == 1 and not self.always:
            return
        _, a, b = next(r for r in self.regions if r[0] == name)
        view = self.g[a:b]
        if self._cuda:
            self._comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm):
                ev = None
                if self.timing:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record(self._comm)
                buf = view.to(torch.bfloat16) if self.wire_bf16 else view
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending[name] = (work, buf, view, ev)
        else:
            buf = view.to(torch.bfloat16) if self.wire_bf16 else view
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending[name] = (work, buf, view, None)

    def finish(self, name: str) -> None:
        """Make the current stream (GPU) / the caller (CPU) wait for region ``name``."""
        item = self._pending.pop(name, None)
        if item is None:
            return
        work, buf, view, ev = item
        if self._cuda:
            # the COMMUNICATION stream waits for the collective (work.wait() orders only the stream that is current when it is
            # called; RCCL runs on a stream of its own), copies a bf16 wire buffer back, and the current stream then waits for it
            with torch.cuda.stream(self._comm):
                work.wait()
                if buf is not view:
                    view.copy_(buf)
                if ev is not None:
                    ev[1].record(self._comm)
                    self._events.append(ev)
            torch.cuda.current_stream().wait_stream(self._comm)
        else:
            work.wait()
            if buf is not view:
                view.copy_(buf)

    def comm_ms(self) -> float:
        """With ``timing`` on: milliseconds between launch and arrival summed over the regions exchanged since the last call (device
        events on the communication stream; synchronises)."""
        if not self._events:
            return 0.0
        torch.cuda.synchronize()
        t = sum(a.elapsed_time(b) for a, b in self._events)
        self._events = []
        return t

    def all(self) -> None:
        self.exchange()

    def exchange(self, late=(), between=None, on_arrival=None) -> None:
        """One step's exchange.  Every region not in ``late`` is launched at once (list order: every rank must issue its collectives
        in the same order); then ``between()`` runs -- the second half of a split backward pass, which WRITES the late regions while
        the early ones travel -- and the late regions follow.  Regions are finished in list order; ``on_arrival(i, name, a, b)``
        is called for each as soon as it has arrived (AdamW of that region).  Without ``between`` nothing is late."""
        late = [n for n in late] if between is not None else []
        for n, _, _ in self.regions:
            if n not in late:
                self.start(n)
        if between is not None:
            between()
            for n, _, _ in self.regions:
                if n in late:
                    self.start(n)
        for i, (n, a, b) in enumerate(self.regions):
            self.finish(n)
            if on_arrival is not None:
                on_arrival(i, n, a, b)


class Pretrainer:
    """One object = the reference's models + AdamW + NT-Xent loop state for one rank."""

    def __init__(self, pc_model, img_model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, temperature=0.1,
                 cmid_weight=1.0, process_group=None, world_size: Optional[int] = None, force_data_parallel: bool = False,
                 loss_scale: Optional[float] = None, growth_interval: int = 2000, growth_factor: float = 2.0, backoff_factor: float = 0.5):
        """loss_scale: GradScaler's init_scale (pretrain.py:154).  None: torch's default 2 ** 16 (or VPF_LOSS_SCALE from the environment:
        the test suite's 2 - 8-pair batches have per-sample gradients 8 - 30 x those of a 64-pair batch and inspect a single step, so
        they start at the scale the scaler would back off to); 0 disables loss scaling and the overflow check (gradients below
        fp16's range are then lost in the backward pass's 16-bit operands)."""
        if loss_scale is None:
            loss_scale = float(os.environ.get("VPF_LOSS_SCALE", "65536"))
        self.pc_model, self.img_model = pc_model, img_model
        self.temperature, self.cmid_weight = temperature, cmid_weight
        self.flat = FlatParams([pc_model, img_model])
        self.group = process_group
        if world_size is None:
            world_size = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.world = world_size
        # the N > 1 code path (split capture, region-wise exchange, AdamW per region) can be forced in a one-rank group: that is how the
        # RCCL path is exercised on a single GPU (tests/test_boundary_gpu.py)
        self.dp = world_size > 1 or force_data_parallel
        dev = self.flat.p.device
        # {lr, beta1, beta2, eps, weight_decay, grad_scale, step, skip | loss_scale, growth_tracker, growth_interval, found_inf,
        #  growth_factor, backoff_factor, skipped_steps, -}  (include/vipformer_hip.h: vpf_adamw_step)
        self.scaled = bool(loss_scale) and L.H16 == torch.float16
        self.hyper = torch.tensor([lr, betas[0], betas[1], eps, weight_decay, 1.0 / world_size, 0.0, 0.0,
                                   float(loss_scale) if self.scaled else 0.0, 0.0, float(growth_interval), 0.0,
                                   float(growth_factor), float(backoff_factor), 0.0, 0.0], dtype=torch.float32, device=dev)
        self.device = dev
        # the image branch is independent of the point-cloud branch until the CMC loss: it runs on its own
        # stream so its kernels fill the CUs that FPS / kNN / the small GEMMs of the pc branch leave idle
        self.overlap = True
        self.fused_losses = True
        # FPS + kNN at the head of the image branch's stream (the point-cloud stream then starts with the K / V producer, which needs only
        # the raw points, and meets the groups in front of Group2Emb) or where the model runs them, at the head of the point-cloud stream.
        # Round 4, alternating runs (NOTES.md): with the K / V producer's backward handed to the other stream FROM THE POINT WHERE ITS
        # INPUT GRADIENT EXISTS (ops.KvBwdDeferral.mark_ready) the first placement wins by 0.07 ms -- the forward join is balanced (both
        # branches within 25 us) and so is the end of backward.
        self.preproc_on_side = os.environ.get("VPF_PREPROC_ON_SIDE", "1") in ("1", "2")
        # "2" (round 6 experiment): FPS + kNN on a THIRD stream -- the image branch starts at once, the K / V producer runs on the
        # point-cloud stream, and the sampler's few workgroups run beside both; Group2Emb meets the groups as before
        self.preproc_own_stream = os.environ.get("VPF_PREPROC_ON_SIDE", "1") == "2"
        self._pre = torch.cuda.Stream(device=dev) if (dev.type == "cuda" and self.preproc_own_stream) else None
        self.main_first = os.environ.get("VPF_MAIN_FIRST", "0") == "1"       # measured: no gain in the unmarked step (4.31 vs 4.29 ms), off
        # experiment: the image branch's encoder held back until Group2Emb's forward is queued on the point-cloud
        # stream, so that the resident image attention does not run beside adapter_kv_fwd / g2e_fwd_a (NOTES round 6 has the A/B)
        self.img_gate = os.environ.get("VPF_IMG_GATE", "0") == "1"
        self.timeline = None                 # an ops.Timeline: device timestamps at the branch boundaries (tools/step_timeline.py); None = no marks
        # optional: the point-cloud branch's grouped weight gradients on the image branch's stream behind its backward (ops.WgradDeferral).
        # Measured +0.11 ms/step on MI355X: the two branches already share the CUs for most of the step, the step is bound by the SUM
        # of kernel time (6.9 ms over 4.55 ms of wall), not by the longer stream
        self.defer_wgrad = os.environ.get("VPF_DEFER_WGRAD", "0") == "1"
        # the K / V producer's backward (weight gradients only, the LAST node autograd runs) on the image branch's stream, beside the
        # small launches at the end of Group2Emb's backward instead of behind them (ops.AdapterKVFn.backward, ops.KvBwdDeferral)
        self.kv_bwd_on_side = os.environ.get("VPF_KV_BWD_ON_SIDE", "1") == "1"
        self._side = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._graph = None
        self._static = None
        self.losses = None
        self.zero_grad_in_optimizer = False      # AdamW leaves the gradient buffer zeroed for the next step (set by capture())
        self._g_clean = False
        # the loss gradient's seed: GradScaler.scale(loss) (pretrain.py:209) -- a VIEW of the device-resident scale, so a replayed graph
        # backpropagates with whatever the last update left there; a persistent 1.0 without loss scaling (no ones_like fill per step)
        self._one = self.hyper[8] if self.scaled else torch.ones((), dtype=torch.float32, device=dev)
        # gradient regions in the order backward finishes them: the image model (second half of the flat buffer; its backward runs
        # on the side stream and ends first), then the point-cloud model (Group2Emb / adapter weight gradients come last)
        n_pc = len({id(p) for p in pc_model.parameters()})
        cut = self.flat.offsets[n_pc] if n_pc < len(self.flat.offsets) else self.flat.numel
        # N > 1 with a captured step: backward runs as two graphs -- everything down to the encoder's inputs, then the input stages
        # (Group2Emb, position MLP, point adapter + the cross-attention K / V projections: ~1.2 of the step's 4.5 ms).  The gradients
        # the first graph completes (the image model, the point-cloud encoder and head: 95 % of the bytes) are on the wire while
        # the second graph runs.  "late" = parameters whose gradient the second graph writes.
        late = ("group2emb.", "position_emb.", "input_adapter.", "encoder.cross_attn_n.0.module.q_norm.", "encoder.cross_attn_n.0.module.kv_norm.",
                "encoder.cross_attn_n.0.module.attention.q_proj.", "encoder.cross_attn_n.0.module.attention.k_proj.",
                "encoder.cross_attn_n.0.module.attention.v_proj.")
        # (q_norm and q_proj are complete after the first graph, but they sit between late parameters in the flat order: counting them
        #  late makes the late set ONE run -- three collectives per step (image, early, late) instead of seven)
        names = [k for k, _ in pc_model.named_parameters()]
        ends = self.flat.offsets[1:n_pc] + [cut]
        runs = []                                             # maximal runs of early / late parameters of the point-cloud model
        for k, a0, b0 in zip(names, self.flat.offsets[:n_pc], ends):
            is_late = k.startswith(late)
            if runs and runs[-1][0] == is_late:
                runs[-1][2] = b0
            else:
                runs.append([is_late, a0, b0])
        self.regions = [("img", cut, self.flat.numel)]
        self.regions += [(f"pc.early{i}", a0, b0) for i, (l, a0, b0) in enumerate(runs) if not l]
        self.late_regions = [(f"pc.late{i}", a0, b0) for i, (l, a0, b0) in enumerate(runs) if l]
        self.regions += self.late_regions
        self.exchange = GradientExchange(self.flat.g, self.regions, world_size, process_group,
                                         wire_bf16=os.environ.get("VPF_GRAD_WIRE", "f32") == "bf16", always=force_data_parallel)
        self.overlap_comm = os.environ.get("VPF_COMM_OVERLAP", "1") == "1"
        self._graph2, self._cut = None, None

    # ------------------------------------------------------------------ pieces
    def broadcast_parameters(self, src: int = 0) -> None:
        """DDP constructor semantics (pretrain.py:104-105): rank 0's parameters and buffers everywhere."""
        if self.dp:
            dist.broadcast(self.flat.p, src, group=self.group)
            for m in (self.pc_model, self.img_model):
                for b in m.buffers():
                    dist.broadcast(b, src, group=self.group)
            self.flat.refresh_shadow()

    def sync_buffers(self, src: int = 0) -> None:
        """DistributedDataParallel(broadcast_buffers=True) re-broadcasts rank 0's BatchNorm running statistics in front of every forward
        pass (pretrain.py:104-105).  Training-mode BatchNorm never reads them, so the step does not pay for that; call this before
        anything that does -- the per-epoch eval-mode probe (pretrain.py:228-276), saving a checkpoint from a rank other than 0."""
        if self.dp:
            sync_module_buffers((self.pc_model, self.img_model), src, self.group)

    def eval(self) -> "Pretrainer":
        """pretrain.py:229-231 (`pc_model_ddp.eval()`, `img_model_ddp.eval()`) for the per-epoch probe: both models to eval mode AND,
        data-parallel, rank 0's BatchNorm running statistics to every rank first -- the reference's DDP wrappers broadcast them in
        front of every forward pass, this trainer only here, where they are read (VERDICT r05 missing 4: nobody has to remember)."""
        self.sync_buffers(0)
        self.pc_model.eval(); self.img_model.eval()
        return self

    def train(self) -> "Pretrainer":
        """pretrain.py:160-162."""
        self.pc_model.train(); self.img_model.train()
        return self

    def forward_backward(self, pc_t1, pc_t2, imgs):
        """pretrain.py:174-209 (modality 'both').  imgs: [b,3,H,W] as the loader yields it."""
        with ops.rng.pinned():       # one dropout state per step, advanced on the device by optimizer_step (graph-replayable)
            return self._forward_backward(pc_t1, pc_t2, imgs)

    def backward_inputs(self, cut) -> None:
        """Second half of a split backward pass: from the gradients that arrived at the encoder's (detached) inputs through Group2Emb,
        the position MLP and the point adapter / K,V producer (see _PointBackbone._cut_here)."""
        with ops.rng.pinned():
            pairs = [(t, d.grad) for t, d in cut if d.grad is not None]
            if pairs:
                torch.autograd.backward([t for t, _ in pairs], [g for _, g in pairs])
            ops.join_wgrad_streams()

    def _forward_backward(self, pc_t1, pc_t2, imgs, cut=None):
        self.flat.attach_grads()
        if not self._g_clean:
            self.flat.g.zero_()                             # pretrain.py:174 (skipped when the last AdamW launch left it zero)
        self._g_clean = False
        imgs = imgs.permute(0, 2, 3, 1)                     # pretrain.py:179 (a view; strides go to the kernel)
        b = pc_t1.shape[0]
        if (pc_t1.is_contiguous() and pc_t2.is_contiguous() and pc_t1.shape == pc_t2.shape
                and pc_t1.untyped_storage().data_ptr() == pc_t2.untyped_storage().data_ptr()
                and pc_t1.data_ptr() + pc_t1.numel() * pc_t1.element_size() == pc_t2.data_ptr()):
            pc = torch.as_strided(pc_t1, (2 * b,) + tuple(pc_t1.shape[1:]), pc_t1.stride(), pc_t1.storage_offset())   # the views ARE cat(t1, t2)
        else:
            pc = torch.cat([pc_t1, pc_t2], dim=0)           # pretrain.py:183
        tl = self.timeline
        stamp = (lambda x, f, b: ops.StampFn.apply(x, tl, f, b)) if tl is not None else (lambda x, f, b: x)
        if tl is not None:
            tl.mark("step.begin")
        if self.overlap and self._side is not None:
            main = torch.cuda.current_stream()
            self._side.wait_stream(main)
            side_out = {}

            def side_branch():
                """FPS + kNN grouping and the whole image forward on the side stream.  Returns the groups for the point-cloud branch."""
                groups = None
                if self._pre is not None:
                    from .model.pointcloud.utils import divide_patches
                    self._pre.wait_stream(main)
                    with torch.cuda.stream(self._pre):
                        nb, ct = divide_patches(pc, self.pc_model.num_groups, self.pc_model.group_size)
                        ev = torch.cuda.Event()
                        ev.record(self._pre)
                        nb.record_stream(main); ct.record_stream(main)
                        groups = (nb, ct, ev)
                with torch.cuda.stream(self._side):
                    if tl is not None:
                        tl.mark("side.begin")
                    if self.preproc_on_side and self._pre is None:
                        # FPS + kNN grouping ahead of the image branch on ITS stream (it has ~0.6 ms of slack): the point-cloud
                        # stream starts with the K / V producer, which needs only the raw points, and meets the groups later
                        from .model.pointcloud.utils import divide_patches
                        nb, ct = divide_patches(pc, self.pc_model.num_groups, self.pc_model.group_size)
                        ev = torch.cuda.Event()
                        ev.record(self._side)
                        nb.record_stream(main); ct.record_stream(main)
                        groups = (nb, ct, ev)
                        if tl is not None:
                            tl.mark("side.preproc.end")
                    if not self.img_gate:
                        side_out["img"] = stamp(self.img_model(imgs)[0], "img.fwd.end", "img.bwd.begin")
                return groups

            def gated_image_branch():
                """VPF_IMG_GATE: the image forward, issued from inside the point-cloud model once Group2Emb's kernels are queued."""
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(self._side):
                    self._side.wait_event(ev)
                    side_out["img"] = stamp(self.img_model(imgs)[0], "img.fwd.end", "img.bwd.begin")

            if tl is not None:
                tl.mark("pc.fwd.begin")
            if self.preproc_on_side and self.main_first:
                # the point-cloud model calls side_branch() itself, BEHIND its K / V producer: in the captured graph the main stream's
                # first kernels are then created before the side stream's ~25 nodes (measured with tools/step_timeline.py: issued the
                # other way round the replayed graph started the point-cloud branch only 168 us into the step)
                feats = stamp(self.pc_model(pc, _groups=side_branch, _cut=cut)[0], "pc.fwd.end", "pc.bwd.begin")
                if "img" not in side_out:
                    (gated_image_branch if self.img_gate else side_branch)()    # (a model path that never asked for the groups)
            else:
                groups = side_branch()
                feats = stamp(self.pc_model(pc, _groups=groups, _cut=cut, _after_g2e=gated_image_branch if self.img_gate else None)[0],
                              "pc.fwd.end", "pc.bwd.begin")
                if "img" not in side_out:
                    gated_image_branch()                            # (a model path that never reached the hook)
            img_feats = side_out["img"]
            main.wait_stream(self._side)
            img_feats.record_stream(main)
        else:
            feats = self.pc_model(pc, _cut=cut)[0]
            img_feats = self.img_model(imgs)[0]
        if self.fused_losses:
            # both NT-Xent losses, the view mean and the weighted sum in three launches (vpf_pretrain_loss_fwd)
            total, parts = ops.pretrain_losses(feats, img_feats, self.temperature, self.cmid_weight)
            loss_imid, loss_cmid = parts[0], parts[1]
        else:
            f1, f2 = feats[:b], feats[b:]
            loss_imid = ops.ntxent_loss(f1, f2, self.temperature)
            loss_cmid = ops.ntxent_loss((f1 + f2) / 2, img_feats, self.temperature)
            total = loss_imid + self.cmid_weight * loss_cmid
        defer = ops.WgradDeferral(self._side) if (self.overlap and self._side is not None and self.defer_wgrad) else None
        ops.cfg.wgrad_defer = defer
        kvd = ops.KvBwdDeferral(self._side) if (self.overlap and self._side is not None and self.kv_bwd_on_side) else None
        ops.cfg.kv_bwd_defer = kvd
        try:
            total.backward(self._one)                       # (a persistent 1.0: no ones_like fill launch per step)
        finally:
            ops.cfg.wgrad_defer = None
            ops.cfg.kv_bwd_defer = None
        if kvd is not None:
            kvd.drain()                                     # the K / V producer's backward, on the image branch's stream
        if defer is not None:
            defer.drain()                                   # the point-cloud branch's grouped weight gradients, behind the image branch's backward
        if tl is not None:
            tl.mark("main.bwd.end")
            if self.overlap and self._side is not None:
                with torch.cuda.stream(self._side):
                    tl.mark("side.bwd.end")
        if self.overlap and self._side is not None:
            # the kernels write weight gradients themselves (autograd sees no leaf accumulation on the side stream and
            # therefore does not join it): the image branch's backward must land before anything reads the gradients
            torch.cuda.current_stream().wait_stream(self._side)
        ops.join_wgrad_streams()                             # side-stream weight gradients land before anything reads them
        return total.detach(), loss_imid.detach(), loss_cmid.detach()

    def allreduce_gradients(self) -> None:
        """The step's exchange: SUM of the flat gradient over ranks (mean folded into AdamW), region by region."""
        self.exchange.all()

    # ------------------------------------------------------------------ GradScaler state (device-resident)
    @property
    def loss_scale(self) -> float:
        """The current loss scale (a device read: not for the hot loop).  1.0 without loss scaling."""
        return float(self.hyper[8]) if self.scaled else 1.0

    @property
    def skipped_steps(self) -> int:
        """Steps the overflow check skipped so far (GradScaler: found_inf)."""
        return int(self.hyper[14])

    def unscaled_grad(self) -> torch.Tensor:
        """flat.g without the loss scale (what GradScaler.unscale_ would leave in .grad); a copy."""
        return self.flat.g / self.hyper[8] if self.scaled else self.flat.g.clone()

    def unscale_(self) -> None:
        """GradScaler.unscale_(optimizer): divide the flat gradient by the loss scale in place (for gradient clipping, or for reading
        ``p.grad`` in the loss's own units); the next optimizer step then takes the gradients as they are."""
        if self.scaled and float(self.hyper[15]) == 0.0:
            self.flat.g.div_(self.hyper[8])
            self.hyper[15] = 1.0

    def _check_grads(self) -> None:
        """scaler.step's inf / NaN check (pretrain.py:210) over the whole flat gradient, on the device: sets hyper[11]."""
        if self.scaled:
            L.call("vpf_grad_check", self.flat.g, self.flat.numel, self.hyper)

    def _adamw_region(self, a: int, b: int, advance: bool) -> None:
        f = self.flat
        L.call("vpf_adamw_step", f.p[a:b], f.g[a:b], f.m[a:b], f.v[a:b], f.s[a:b], b - a, self.hyper,
               int(advance) | (2 if self.zero_grad_in_optimizer else 0))

    def exchange_and_step(self, between=None) -> None:
        """N > 1: regions are reduced asynchronously on the communication stream; AdamW of a region runs as soon as it has
        arrived (the image region's update overlaps the point-cloud region's transfer).  between: the second half of a split
        backward pass, launched after the early regions' transfers and before the late regions'.  The bias-correction step counter advances
        with the last region; the dropout state once per step."""
        last = len(self.regions) - 1
        if self.scaled:
            # the skip decision needs EVERY region's reduced gradient (an inf survives the SUM, so all ranks decide alike): AdamW runs
            # once, behind the check, instead of region by region on arrival
            self.exchange.exchange([n for n, _, _ in self.late_regions], between, None)
            self._check_grads()
            self._adamw_region(0, self.flat.numel, True)
        else:
            self.exchange.exchange([n for n, _, _ in self.late_regions], between,
                                   lambda i, n, a, b: self._adamw_region(a, b, i == last))
        self._g_clean = self.zero_grad_in_optimizer
        ops.rng.advance(self.device)

    def set_lr(self, lr: float) -> None:
        """The learning-rate schedule's hook (the reference steps a cosine / warm-restart schedule per epoch, pretrain.py:136-142,
        311): writes the device-resident hyper-parameter the AdamW kernel reads, so it also takes effect in a captured graph."""
        self.hyper[0] = float(lr)

    def optimizer_step(self) -> None:
        f = self.flat
        self._check_grads()
        L.call("vpf_adamw_step", f.p, f.g, f.m, f.v, f.s, f.numel, self.hyper, 1 | (2 if self.zero_grad_in_optimizer else 0))
        self._g_clean = self.zero_grad_in_optimizer
        ops.rng.advance(self.device)

    # ------------------------------------------------------------------ whole step
    def step(self, pc_t1, pc_t2, imgs):
        losses = self.forward_backward(pc_t1, pc_t2, imgs)
        if self.dp:
            self.exchange_and_step()
        else:
            self.optimizer_step()
        self.losses = losses
        return losses

    def capture(self, pc_t1, pc_t2, imgs, warmup: int = 3, keep_grads: bool = False):
        """Capture forward+backward (+AdamW when single-rank) into a hipGraph on static input buffers.
        Returns the static (pc_t1, pc_t2, imgs) tensors to copy new batches into.  The two views live in ONE buffer (cat(t1, t2) of
        pretrain.py:183 is then a view, not a copy per step); unless keep_grads, AdamW leaves the flat gradient zeroed for the next
        replay (no 33 MB fill per step) -- flat.g then reads zero after a step."""
        both = torch.cat([pc_t1, pc_t2], dim=0).contiguous()
        b = pc_t1.shape[0]
        self._static = (both[:b], both[b:], imgs.clone())
        self.zero_grad_in_optimizer = not keep_grads
        # warm-up (lazy allocations, packed-weight buffers, func attributes) must not train: AdamW runs with its skip flag
        # (parameters, moments and the bias-correction step stay put), BatchNorm buffers and the dropout step are restored
        bufs = [b for m in (self.pc_model, self.img_model) for b in m.buffers()]
        keep = [b.clone() for b in bufs]
        rng_keep = ops.rng.state(self.device).clone()
        skip_keep = self.hyper[7].clone()
        self.hyper[7] = 1.0
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.step(*self._static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.hyper[7] = skip_keep
        self.hyper[11] = 0.0                                   # (an overflow seen by a warm-up step is not the first real step's business)
        ops.rng.state(self.device).copy_(rng_keep)
        with torch.no_grad():
            for b, k in zip(bufs, keep):
                b.copy_(k)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        split = self.dp and self.overlap_comm
        self._cut = [] if split else None
        # with a process group alive, RCCL's watchdog thread polls events while we capture: only THIS thread's calls may be checked
        mode = "thread_local" if (self.dp or (dist.is_available() and dist.is_initialized())) else "global"
        with torch.cuda.graph(self._graph, stream=side, capture_error_mode=mode):   # (the warm-up's stream: per-stream scratch buffers exist already)
            with ops.rng.pinned():
                self.losses = self._forward_backward(*self._static, cut=self._cut)
            if not self.dp:
                self.optimizer_step()
        self._graph2 = None
        if split and self._cut:
            self._graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph2, stream=side, pool=self._graph.pool(), capture_error_mode=mode):
                self.backward_inputs(self._cut)
        return self._static

    def replay(self):
        """One captured step (inputs are whatever the static buffers hold)."""
        self._graph.replay()
        if self.dp:
            self.exchange_and_step(self._graph2.replay if self._graph2 is not None else None)
        return self.losses


class GraphedStep:
    """A training step of the caller's own -- the fine-tune loops of the reference have no trainer class (ft_partseg.py:145-176,
    ft_cls.py: zero_grad, forward, loss, backward, clip_grad_norm_, optimizer.step) -- captured into ONE hipGraph and replayed.

        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)     # the optimizer must be capturable
        def step():                                                               # reads tensors the caller owns and refills in place
            opt.zero_grad(set_to_none=True)
            loss = criterion(model(points, onehot), target)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 10)
            opt.step()
            out["loss"] = loss
        run = GraphedStep(step)          # `warmup` eager steps on a side stream (they DO train), then the capture
        for batch in loader:
            points.copy_(batch.points); ...
            run()

    Nothing in the step may read a device value on the host, and no tensor that still carries the autograd graph of an EARLIER eager
    step (a kept `loss`) may be alive at capture time: its AccumulateGrad nodes are bound to that step's stream.  Dropout masks change from replay to replay: every mask-drawing
    Function snapshots the device-resident state and advances it on the device (ops._Rng), and those launches are part of the graph;
    `torch.randint` of farthest-point sampling is graph-safe in torch.  At 16 clouds x 1 024 points the replayed step of
    CrossFormer_partseg takes 4.8 ms against 8.7 ms eager (the eager step is bound by Python launching ~500 kernels).
    Every replay bumps the process-wide ops._OPT_EPOCH (the replayed optimizer moved parameters without Python noticing): the h16
    weight shadows of EVERY model in the process are then stale by construction, so trainer-managed (ManagedFlat) parameters used
    eagerly in the same process pay one cast launch per parameter on their next forward -- correct, but not free."""

    def __init__(self, step, warmup: int = 3):
        self.step = step
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                step()
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        # The kernels' h16 weight copies (ops.shadow) are cached on (p._version, count of optimizer steps Python has seen).  The
        # captured step must contain the re-casts -- every replay starts from the fp32 weights the previous replay's optimizer wrote --
        # so every cached copy is made stale before the capture (also with warmup = 0 on a model that has run before) ...
        ops._OPT_EPOCH[0] += 1
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            step()
        ops._OPT_EPOCH[0] += 1

    def __call__(self) -> None:
        self.graph.replay()
        # ... and a replay steps the optimizer on the device without Python's optimizer hooks running: the copies the replay left
        # behind hold the weights from BEFORE its update, so an eager forward after it (the evaluation between epochs) must re-cast
        # (without this, every eager forward after the first one computed with weights one step stale).
        ops._OPT_EPOCH[0] += 1


class HostFeeder:
    """The reference copies every batch to the device synchronously at the top of the step (pretrain.py:177: 40 MB of fp32 per 64
    pairs, ~1.3 ms of PCIe in front of a 4.5 ms step).  Here the NEXT batch travels on a copy stream into staging buffers while the
    current step runs; at the top of the next step a device-to-device copy (40 MB at HBM speed: ~10 us) moves it into the captured
    graph's static inputs.  Host tensors must be pinned.

        feeder = HostFeeder(trainer)            # after trainer.capture(...)
        feeder.submit(t1_h, t2_h, imgs_h)       # batch 0
        for next_batch in loader:
            losses = feeder.step()              # consumes what was submitted
            feeder.submit(*next_batch)          # travels while that step runs
    """

    def __init__(self, trainer: "Pretrainer"):
        if trainer._static is None:
            raise L.VpfError("HostFeeder needs a captured trainer (Pretrainer.capture)")
        self.tr = trainer
        self.stage = tuple(torch.empty_like(t) for t in trainer._static)
        self.copy_stream = torch.cuda.Stream(device=trainer.device)
        self.filled = torch.cuda.Event()
        self.consumed = torch.cuda.Event()
        self.consumed.record(torch.cuda.current_stream())
        self._pending = False
        self._raw = None

    def submit(self, pc_t1, pc_t2, imgs) -> None:
        for t in (pc_t1, pc_t2, imgs):
            if t.device.type != "cpu" or not t.is_pinned():
                raise L.VpfError("HostFeeder.submit takes pinned host tensors")
        self.copy_stream.wait_event(self.consumed)            # the staging buffers are free once the last step has copied them out
        with torch.cuda.stream(self.copy_stream):
            for dst, src in zip(self.stage, (pc_t1, pc_t2, imgs)):
                dst.copy_(src, non_blocking=True)
            self.filled.record(self.copy_stream)
        self._pending = True

    def submit_raw(self, raw_clouds, imgs_u8) -> None:
        """The DataLoader workers' augmentations moved to the GPU (vipformer_amd.augment): the host ships ONE raw cloud per pair
        (fp32 [b,N,3]) and the decoded uint8 image ([b,H,W,3]); step() draws the two views (trans_1 twice) and normalises / flips the
        image on the device.  10 MB instead of 40 MB per 64 pairs."""
        for t in (raw_clouds, imgs_u8):
            if t.device.type != "cpu" or not t.is_pinned():
                raise L.VpfError("HostFeeder.submit_raw takes pinned host tensors")
        if self._raw is None or self._raw[0].shape != raw_clouds.shape or self._raw[1].shape != imgs_u8.shape:
            self._raw = (torch.empty(raw_clouds.shape, dtype=torch.float32, device=self.tr.device),
                         torch.empty(imgs_u8.shape, dtype=torch.uint8, device=self.tr.device))
        self.copy_stream.wait_event(self.consumed)
        with torch.cuda.stream(self.copy_stream):
            self._raw[0].copy_(raw_clouds, non_blocking=True)
            self._raw[1].copy_(imgs_u8, non_blocking=True)
            self.filled.record(self.copy_stream)
        self._pending = "raw"

    def step(self):
        if not self._pending:
            raise L.VpfError("HostFeeder.step without a submitted batch")
        cur = torch.cuda.current_stream()
        cur.wait_event(self.filled)
        if self._pending == "raw":
            from . import augment as G
            st = self.tr._static
            st[0].copy_(G.augment_points(self._raw[0])); st[1].copy_(G.augment_points(self._raw[0]))
            st[2].copy_(G.image_u8_normalize(self._raw[1]))
        else:
            for dst, src in zip(self.tr._static, self.stage):
                dst.copy_(src, non_blocking=True)
        self.consumed.record(cur)
        self._pending = False
        return self.tr.replay()


def build_models(D=256, H=4, G=96, K=32, S=6, MR=2, N=1024, img=224, patch=16, atten_drop=0.1, mlp_drop=0.5, n_ca=1,
                 point_channels=3, device="cuda"):
    """utils.py:119-149 (build_model, --mp branch) with the architecture flags spelled out."""
    from .model.pointcloud import CrossFormer_img_mp, CrossFormer_pc_mp, PointCloudInputAdapter
    adapter = PointCloudInputAdapter(pointcloud_shape=(N, point_channels), num_input_channels=D)
    pc = CrossFormer_pc_mp(input_adapter=adapter, num_latents=G, num_latent_channels=D, group_size=K,
                           num_cross_attention_layers=n_ca, num_cross_attention_heads=H, num_self_attention_layers=S,
                           num_self_attention_heads=H, mlp_widen_factor=MR, max_dpr=0.0, atten_drop=atten_drop,
                           mlp_drop=mlp_drop, modal_prior=True)
    im = CrossFormer_img_mp(img_height=img, img_width=img, patch_size=patch, num_latent_channels=D,
                            num_cross_attention_layers=n_ca, num_cross_attention_heads=H, num_self_attention_layers=S,
                            num_self_attention_heads=H, mlp_widen_factor=MR, max_dpr=0.0, atten_drop=atten_drop,
                            mlp_drop=mlp_drop, modal_prior=True)
    return pc.to(device), im.to(device)
