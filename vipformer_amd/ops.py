"""Host-side glue between the mirrored nn.Modules and the C ABI (libvipformer_hip.so).

Everything numerical happens in the HIP kernels; this file owns buffers, autograd wiring
(torch.autograd.Function with hand-written backward sequences) and the small amount of
parameter bookkeeping the fused kernels need:

  * h16 "shadow" copies of fp32 master weights (MFMA operands), refreshed when the
    parameter changes, or owned by the trainer's fused AdamW (which writes them directly);
  * packed q/k/v weights and gradients (one [3D,D] GEMM instead of three);
  * weight gradients: the wgrad kernels accumulate with fp32 atomics (split over the token
    dimension) into a buffer chosen per parameter by ``_GradSink`` -- the trainer's flat
    gradient buffer for parameters a ``Pretrainer`` owns (the Function then returns None:
    nothing is re-added by autograd), a zeroed temporary RETURNED through autograd for every
    other parameter, so ``p.grad`` is populated the standard way and DistributedDataParallel's
    reducer hooks fire (pretrain.py:104-105,209).

Precision contract: fp32 master weights, fp32 residual stream, h16 MFMA operands with fp32
accumulation, h16 activations between fused ops, fp32 statistics (LayerNorm/BatchNorm/softmax).
"""
from __future__ import annotations

import ctypes
import functools
import os
import threading
import weakref
import zlib
from typing import Optional, Sequence

import torch

from . import _lib as L

H16 = L.H16              # the library's 16-bit operand dtype (fp16; _lib.H16)
F32 = torch.float32

EPI_STORE, EPI_GELU, EPI_DROP_RES, EPI_GELU_BWD, EPI_ATOMIC, EPI_RELU, EPI_GROUPBIAS = range(7)


class OpsConfig:
    """Every path switch of this module in ONE object (``ops.cfg``), read once at import; A/B tests and tools flip attributes.
    None of them is a user-facing option: the defaults are the measured-fastest parity-green paths (DESIGN.md sections 4, 5).
    The library's own launch-time knobs live in its VpfDebug struct (``_lib.debug_get`` / ``debug_set``)."""
    __slots__ = ("wgrad_async", "wgrad_group", "wgrad_group_async", "wgrad_deterministic", "wgrad_defer", "sa_debug", "sa_split_attn",
                 "sa_fused_bwd", "sa_fused", "enc_bwd_hook", "adapter_kv_fused", "adapter_kv_bwd_fused", "enc_fused", "g2e_bn_merged",
                 "g2e_conv1_bwd_fused", "ca_front_fused", "wgrad_stack", "ca_front_bwd_fused", "ca_kv_bwd_fused", "pgrad_flush", "kv_bwd_defer")

    def __init__(self, env=os.environ):
        self.wgrad_async = False          # weight-gradient GEMMs on a side stream: measured slower (cross-stream event cost > overlap gain)
        self.wgrad_group = True           # one grouped launch per layer for the weight gradients (vpf_wgrad_group)
        self.wgrad_group_async = False    # issue the grouped launches on a side stream (joined by join_wgrad_streams)
        self.wgrad_deterministic = False  # grouped weight gradients through the split-K workspace instead of fp32 atomics (slower, bitwise reproducible)
        self.wgrad_defer = None           # the open WgradDeferral, if any (process-wide: autograd runs backward on its own thread)
        self.sa_debug = None              # int64 tensor of >= 8: phase cycle counters of workgroup 0 of the last fused layer launch
        self.sa_split_attn = None         # None: per-shape default; True: vpf_attention_fwd + fused tail; False: attention inside the layer kernel
        self.sa_fused_bwd = True          # backward of the stack through vpf_sa_layer_bwd_mlp / _qkv instead of the block-by-block kernels
        self.sa_fused = True              # Encoder.forward uses the one-kernel-per-layer path when the shapes allow it
        self.enc_bwd_hook = None          # schedule experiments (tools/step_timeline.py): callable(cross_attention_layer, i)
        self.adapter_kv_fused = env.get("VPF_ADAPTER_KV", "1") == "1"      # point adapter + kv LayerNorm + K / V projections as one kernel
        self.adapter_kv_bwd_fused = True
        self.enc_fused = True             # cross-attention layer tail fused as well (EncoderFusedFn) when the shapes allow it
        self.ca_front_fused = env.get("VPF_CA_FRONT", "1") == "1"     # position MLP + (tokens + pos) + q_norm + q projection of the point-cloud branch as one kernel
        self.ca_front_bwd_fused = env.get("VPF_CA_FRONT_BWD", "1") == "1"      # the cross-attention layer's query-side backward as one row-block kernel (vpf_ca_front_bwd)
        self.ca_kv_bwd_fused = env.get("VPF_CA_KV_BWD", "0") == "1"            # ... and its key / value side where the kv input is an f32 tensor (image branch: vpf_ca_kv_bwd; measured: no gain, 3.93 vs 3.94 ms: off)
        self.kv_bwd_defer = None          # the open KvBwdDeferral, if any: the K / V producer's backward (weight gradients only) is issued by its drain()
        self.pgrad_flush = 32             # LayerNorm parameter-gradient folds queued per vpf_ln_pgrad_reduce launch (tests lower it: ADVICE r03)
        self.wgrad_stack = env.get("VPF_WGRAD_STACK", "1") == "1"               # the weight gradients of a whole fused encoder stack as ONE grouped launch at the end of its backward
        self.g2e_bn_merged = env.get("VPF_G2E_BN_MERGED", "1") != "0"          # BatchNorm bookkeeping of Group2Emb as single launches
        self.g2e_conv1_bwd_fused = env.get("VPF_G2E_CONV1_FUSED", "1") == "1"  # conv2 dgrad inside the first conv's backward (tests run both)


cfg = OpsConfig()


# --------------------------------------------------------------------------- RNG state
class _Rng:
    """Device-resident dropout state {seed_lo, seed_hi, step, 0}.  A keep decision is a pure function of (state, site,
    element index), so backward regenerates forward's mask from the state the forward pass used.

    Two regimes:
      * default: every training-mode autograd Function that draws masks takes a SNAPSHOT of the state (``acquire``) and bumps the
        process state behind it, keeps the snapshot for its backward pass -- consecutive forward passes (and two calls of the same
        module inside one forward pass) therefore never share masks, whoever drives the loop (pretrain.py's own loop body, a
        torch optimizer, DistributedDataParallel ...);
      * ``with rng.pinned():`` (the Pretrainer, and tests that export the kernels' masks): every Function draws from the process
        state itself and nobody bumps it implicitly -- the owner calls ``advance`` once per step, on the device, so a captured
        hipGraph replays with fresh masks and costs no extra launches."""

    def __init__(self):
        self.states = {}
        self.seed_value = 0x1234ABCD5678EF01
        self._pin = 0

    @staticmethod
    def _index(device) -> int:
        if torch.device(device).type != "cuda":
            raise L.VpfError("vipformer_amd ops run on an MI355X only (got a CPU tensor); there is no CPU fallback")
        idx = torch.device(device).index
        return torch.cuda.current_device() if idx is None else idx

    def state(self, device) -> torch.Tensor:
        key = self._index(device)
        st = self.states.get(key)
        if st is None:
            lo, hi = self.seed_value & 0xFFFFFFFF, (self.seed_value >> 32) & 0xFFFFFFFF
            to_i32 = lambda v: v - (1 << 32) if v >= (1 << 31) else v
            st = torch.tensor([to_i32(lo), to_i32(hi), 0, 0], dtype=torch.int32, device=f"cuda:{key}")
            self.states[key] = st
        return st

    def seed(self, seed: int) -> None:
        self.seed_value = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.states.clear()

    def advance(self, device) -> None:
        L.call("vpf_rng_advance", self.state(device))

    def acquire(self, device, training: bool = True) -> torch.Tensor:
        """The state a Function draws its masks from (and keeps for its backward pass)."""
        st = self.state(device)
        if self._pin or not training:
            return st
        snap = st.clone()
        L.call("vpf_rng_advance", st)
        return snap

    def pinned(self):
        return _Pinned(self)


class _Pinned:
    def __init__(self, r):
        self.r = r

    def __enter__(self):
        self.r._pin += 1
        return self.r

    def __exit__(self, *exc):
        self.r._pin -= 1


rng = _Rng()
_site_counter = [0]


def new_site() -> int:
    """A dropout site id that is unique in the process (modules built on their own).  Top-level models overwrite the sites of
    their sub-modules with ``assign_sites`` so that masks do not depend on how many modules were built before."""
    _site_counter[0] += 1
    return _site_counter[0]


def assign_sites(model: torch.nn.Module, salt: str) -> None:
    """Deterministic dropout sites: crc32 of (salt, qualified module name).  Independent of construction order and of every
    other model in the process; ``salt`` keeps the point-cloud and the image branch (same module names) on different masks."""
    for name, m in model.named_modules():
        for attr in ("site", "site_attn"):
            if hasattr(m, attr):
                setattr(m, attr, zlib.crc32(f"{salt}:{name}:{attr}".encode()) & 0xFFFFFFFF)


def dropout_keep_mask(site: int, p: float, shape, device, state: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The keep mask (uint8) the kernels use for ``site`` at ``state`` (default: the process state -- what the Functions draw
    from inside ``rng.pinned()``).  Tests."""
    n = 1
    for s in shape:
        n *= int(s)
    out = torch.empty(n, dtype=torch.uint8, device=device)
    L.call("vpf_dropout_mask", out, n, state if state is not None else rng.state(device), site, float(p))
    return out.view(*shape)


# --------------------------------------------------------------------------- parameter packing / shadows / gradient sinks
class ManagedFlat:
    """What a trainer registers for parameters it owns (train.FlatParams): flat fp32 values, flat fp32 gradients the weight-
    gradient kernels accumulate into DIRECTLY, and the h16 shadow its fused AdamW rewrites.  Parameters carry a weak reference
    (``p._vpf_managed``) -- no process-wide registry, two trainers in one process never alias, a dead trainer costs nothing."""

    def __init__(self, flat_f32: torch.Tensor, flat_h16: torch.Tensor, flat_grad: Optional[torch.Tensor] = None):
        self.p, self.s, self.g = flat_f32, flat_h16, flat_grad

    def grad_view(self, param: torch.nn.Parameter, offset: int) -> torch.Tensor:
        """``param.grad`` as the view of the flat gradient buffer it must be -- re-installed if somebody dropped it
        (optimizer.zero_grad(set_to_none=True), model.zero_grad()) or replaced it."""
        g = param.grad
        if g is None or self.g is None or g.data_ptr() != self.g.data_ptr() + 4 * offset:
            if self.g is None:
                if g is None:
                    param.grad = torch.zeros_like(param.data)
                return param.grad
            param.grad = self.g[offset:offset + param.numel()].view_as(param.data)
        return param.grad

    def adopt(self, param: torch.nn.Parameter, offset: int) -> None:
        param._vpf_managed = (weakref.ref(self), offset)
        param._vpf_ver = (param._version, _OPT_EPOCH[0])

    def recast(self, param: torch.nn.Parameter, offset: int) -> None:
        n = param.numel()
        L.call("vpf_cast_f32_h16", self.p[offset:offset + n], self.s[offset:offset + n], n)
        param._vpf_ver = (param._version, _OPT_EPOCH[0])


def _managed(p):
    m = getattr(p, "_vpf_managed", None)
    if m is None:
        return None, 0
    owner = m[0]()
    if owner is None or p.data_ptr() != owner.p.data_ptr() + 4 * m[1]:       # trainer gone, or the parameter was re-pointed (.to())
        return None, 0
    return owner, m[1]


def clear_managed_shadows() -> None:
    """Kept for callers of the round-1 API: ownership now lives on the parameters (weak references), nothing to clear."""


def _adjacent(ts: Sequence[torch.Tensor]) -> bool:
    for a, b in zip(ts[:-1], ts[1:]):
        if not a.is_contiguous() or not b.is_contiguous() or a.data_ptr() + a.numel() * a.element_size() != b.data_ptr():
            return False
        if a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr():
            return False
    return True


def pack_params(params: Sequence[torch.nn.Parameter]) -> None:
    """Make the fp32 storage of ``params`` adjacent (in order) so they can be used as one matrix."""
    if len(params) == 1 or _adjacent([p.data for p in params]):
        return
    flat = torch.cat([p.data.reshape(-1) for p in params])
    off = 0
    for p in params:
        p.data = flat[off:off + p.numel()].view_as(p.data)
        off += p.numel()


class _GradSink:
    """Where the weight-gradient kernels of ONE Function.backward write.

    * parameters owned by a trainer (``ManagedFlat``): straight into the trainer's flat gradient buffer (``p.grad`` is a view of
      it); the Function returns None for them -- nothing is re-added by autograd;
    * every other parameter: into a zeroed temporary that the Function RETURNS as that input's gradient, so autograd's
      AccumulateGrad runs -- ``p.grad`` is populated the standard way and DistributedDataParallel's reducer hooks fire
      (pretrain.py:104-105,209)."""

    def __init__(self, params):
        self.params = params
        self.tmp = {}

    def buf(self, p):
        owner, off = _managed(p)
        if owner is not None:
            return owner.grad_view(p, off)
        t = self.tmp.get(id(p))
        if t is None:
            t = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
            self.tmp[id(p)] = t
        return t

    def packed(self, params):
        owners = [_managed(p) for p in params]
        if all(o is not None for o, _ in owners):
            grads = [o.grad_view(p, off) for p, (o, off) in zip(params, owners)]
            if _adjacent(grads):
                n = sum(p.numel() for p in params)
                g0 = grads[0]
                return torch.as_strided(g0, (n,), (1,), g0.storage_offset()) if len(params) > 1 else g0.view(-1)
        hit = [self.tmp.get(id(p)) for p in params]
        if all(h is not None for h in hit) and _adjacent(hit):
            n = sum(p.numel() for p in params)
            return torch.as_strided(hit[0], (n,), (1,), hit[0].storage_offset()) if len(params) > 1 else hit[0].view(-1)
        if any(h is not None for h in hit):
            raise L.VpfError("packed gradient requested after a separate gradient buffer of the same parameters")
        flat = torch.zeros(sum(p.numel() for p in params), dtype=F32, device=params[0].device)
        off = 0
        for p in params:
            self.tmp[id(p)] = flat[off:off + p.numel()].view_as(p.data)
            off += p.numel()
        return flat

    def collect(self):
        return tuple(self.tmp.get(id(p)) for p in self.params)


_tls = threading.local()
_tls_scale = threading.local()


def _sink_stack():
    st = getattr(_tls, "stack", None)
    if st is None:
        st = _tls.stack = []
    return st


# ---- internal gradient scale: fp16 gradient operands for loops that have no GradScaler (ft_partseg.py:145-176)
# The pre-training loop and ft_cls.py scale their loss (pretrain.py:154,209; ft_cls.py:146,166,179-181: autocast + GradScaler); the
# part-segmentation fine-tune loop (ft_partseg.py) runs in fp32 without a scaler.  With fp16
# gradient operands their backward pass loses what lies below 6e-5 (a CrossEntropyLoss averaged over 16 x 1024 points hands the head
# gradients of 1e-6: measured, tools/diag_ft_scale.py: all-parameter cosine 0.964 against the same step with a scaled loss).  The
# fine-tune MODELS therefore scale inside: their forward runs under ``internal_grad_scale()`` and returns its logits through
# ScaleGradFn, which -- in the backward pass -- multiplies the incoming gradient by S = 2^k, k = floor(log2(1 / max|g|)) clamped to
# [0, 24]: the gradient ENTERS the model with its largest element in [0.5, 1], the regime every parity test of the backward kernels
# runs in, whatever the loss's normalisation.  Every op between the Functions is linear in the gradient, so the whole backward pass of
# the model works on S x the gradient, and every Function created under the context multiplies its PARAMETER gradients by 1 / S before
# autograd sees them.  S is a power of two computed ON THE DEVICE (no host synchronisation, capturable): nothing is rounded by it.  It
# never scales DOWN: under a GradScaler (incoming gradients already large: ft_cls.py) S = 1 and nothing changes.
# What this path does NOT have is a GradScaler's overflow guard: S is chosen from the gradient that ENTERS the model, and a later
# stage that amplifies it beyond fp16's range (BatchNorm backward multiplies by gamma * rstd, rstd up to 316 on a near-constant
# channel) produces inf / NaN parameter gradients that nothing skips -- the reference's fp32 loop has no such cliff.  Callers that
# want the guard run the loop under torch's GradScaler (then S = 1 and the scaler's skip logic applies), as ft_cls.py does.
# Input gradients (pts.requires_grad) leave the model multiplied by S: ScaleGradFn sits at the OUTPUT, nothing at the input divides it
# back out (no shipped loop differentiates with respect to the points).
# Parameters owned by a trainer's flat buffer (ManagedFlat) receive their gradient directly from the kernels, not through the sink:
# they would accumulate S x the gradient, so _sinked refuses that combination loudly.
class _ScaleHolder:
    __slots__ = ("inv", "on")

    def __init__(self, on: bool):
        self.on, self.inv = on, None        # inv: 0-dim device tensor 1 / S, set by ScaleGradFn.backward (the first node of the model's backward pass)


def _current_iscale():
    return getattr(_tls_scale, "h", None)


class internal_grad_scale:
    """``with internal_grad_scale() as h:`` -- Functions created inside hand back parameter gradients multiplied by ``h.inv`` (see above)."""

    def __init__(self, on: bool = True):
        self.h = _ScaleHolder(bool(on) and L.H16 == torch.float16)

    def __enter__(self):
        self.prev = _current_iscale()
        _tls_scale.h = self.h if self.h.on else None
        return self.h

    def __exit__(self, *exc):
        _tls_scale.h = self.prev


class ScaleGradFn(torch.autograd.Function):
    """Identity; the gradient passing back through it is multiplied by a power of two that brings its largest element into [0.5, 1]
    (never below 1: see internal_grad_scale).  ``holder.inv`` receives the reciprocal for the parameter-gradient sinks."""

    @staticmethod
    def forward(ctx, x, holder):
        ctx.holder = holder
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        amax = g.detach().abs().amax().float().clamp_min(1e-30)
        k = torch.floor(torch.log2(1.0 / amax))
        k = torch.nan_to_num(k, nan=0.0, posinf=24.0, neginf=0.0).clamp_(0.0, 24.0)
        s = torch.exp2(k)
        ctx.holder.inv = 1.0 / s
        return g * s.to(g.dtype), None


def _sinked(backward):
    """Decorator of Function.backward: the trailing ``len(ctx.params)`` entries of the returned tuple (the parameters' slots)
    are replaced by what the weight-gradient kernels wrote for unmanaged parameters (see _GradSink), divided by the internal
    gradient scale the Function's forward ran under (``ctx._vpf_iscale``, set by _scale_aware_forwards)."""

    @functools.wraps(backward)
    def wrapped(ctx, *grads):
        sk = _GradSink(ctx.params)
        stack = _sink_stack()
        stack.append(sk)
        try:
            out = backward(ctx, *grads)
        finally:
            stack.pop()
        n = len(ctx.params)
        if n == 0:
            return out
        got = sk.collect()
        h = getattr(ctx, "_vpf_iscale", None)
        if h is not None and h.inv is not None:
            if any(_managed(p)[0] is not None for p in ctx.params):
                raise L.VpfError("internal_grad_scale: a parameter of this Function is owned by a trainer's flat gradient buffer, which "
                                 "the kernels write directly -- the 1 / S correction cannot reach it.  Train fine-tune models with a torch "
                                 "optimizer (or under a GradScaler with internal_grad_scale = False)")
            seen, ts = set(), []
            for t in got:
                if t is not None:
                    base = t._base if t._base is not None else t              # (packed buffers: several parameters, one allocation)
                    if id(base) not in seen:
                        seen.add(id(base)); ts.append(base)
            if ts:
                torch._foreach_mul_(ts, h.inv)
        return tuple(out[:len(out) - n]) + got

    wrapped._vpf_sinked = True
    return wrapped


def _scale_aware_forwards(namespace) -> None:
    """Every Function of ``namespace`` whose backward is @_sinked records the internal gradient scale its forward ran under."""
    for cls in list(namespace.values()):
        if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and getattr(cls.__dict__.get("backward", None), "__func__", None) is not None:
            bw = cls.__dict__["backward"].__func__
            if not getattr(bw, "_vpf_sinked", False) or getattr(cls, "_vpf_scale_aware", False):
                continue

            def make(f):
                @functools.wraps(f)
                def fwd(ctx, *a, **k):
                    ctx._vpf_iscale = _current_iscale()
                    return f(ctx, *a, **k)
                return fwd

            cls.forward = staticmethod(make(cls.__dict__["forward"].__func__))
            cls._vpf_scale_aware = True


def packed_grad(params: Sequence[torch.nn.Parameter]) -> torch.Tensor:
    """fp32 gradient buffer covering ``params`` contiguously (inside a Function.backward: see _GradSink)."""
    stack = _sink_stack()
    if stack:
        return stack[-1].packed(list(params))
    grads = [p.grad for p in params]
    if all(g is not None for g in grads) and _adjacent(grads):
        n = sum(p.numel() for p in params)
        g0 = grads[0]
        return torch.as_strided(g0, (n,), (1,), g0.storage_offset()) if len(params) > 1 else g0.view(-1)
    flat = torch.zeros(sum(p.numel() for p in params), dtype=F32, device=params[0].device)
    off = 0
    for p, g in zip(params, grads):
        v = flat[off:off + p.numel()].view_as(p.data)
        if g is not None:
            v.copy_(g)
        p.grad = v
        off += p.numel()
    return flat


def grad_buf(p: torch.nn.Parameter) -> torch.Tensor:
    """The fp32 buffer the weight-gradient kernels accumulate into for ``p`` (inside a Function.backward: see _GradSink;
    outside -- direct kernel tests -- ``p.grad``)."""
    stack = _sink_stack()
    if stack:
        return stack[-1].buf(p)
    if p.grad is None:
        p.grad = torch.zeros_like(p.data)
    return p.grad


_OPT_EPOCH = [0]


def _optimizer_stepped(optimizer, args, kwargs) -> None:
    _OPT_EPOCH[0] += 1


try:                                                        # (torch >= 2.1: a process-wide hook behind every Optimizer.step)
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_hook
    _reg_post_hook(_optimizer_stepped)
except Exception:                                           # noqa: BLE001 -- without the hook the `_version` key is all there is
    pass


def shadow(params: Sequence[torch.nn.Parameter]) -> torch.Tensor:
    """h16 copy of the (adjacent) parameters as one flat tensor.  Trainer-owned parameters: a slice of the trainer's shadow
    (rewritten by its fused AdamW), re-cast here if somebody else wrote the parameter since (load_state_dict, an external
    optimizer: anything that bumps ``p._version``; raw ``p.data`` writes do not -- call FlatParams.refresh_shadow() then).
    Other parameters: a cached cast keyed on (version, pointer)."""
    p0 = params[0]
    n = sum(p.numel() for p in params)
    owner, off = _managed(p0)
    if owner is not None:
        end = off
        ok = True
        for p in params:
            o2, f2 = _managed(p)
            if o2 is not owner or f2 != end:
                ok = False
                break
            if (p._version, _OPT_EPOCH[0]) != p._vpf_ver:        # (a torch optimizer stepped, or somebody wrote the parameter in place)
                owner.recast(p, f2)
            end = f2 + p.numel()
        if ok:
            return owner.s[off:off + n]
    # cached on the first parameter OBJECT (dies with it; a (pointer, size) key could alias a freed model).  The key also carries the
    # count of optimizer steps taken in this process (_OPT_EPOCH): torch's FUSED optimizers update the parameters without bumping
    # `_version`, and a model trained with one would otherwise compute with the h16 copy of its initial weights for ever.
    ver = (_OPT_EPOCH[0],) + tuple((p._version, p.data_ptr()) for p in params)
    hit = getattr(p0, "_vpf_shadow", None)
    if hit is not None and hit[0] == ver and hit[1].numel() == n:
        return hit[1]
    src = torch.as_strided(p0.data, (n,), (1,), p0.data.storage_offset()) if len(params) > 1 else p0.data.reshape(-1)
    out = hit[1] if (hit is not None and hit[1].numel() == n and hit[1].device == p0.device) else torch.empty(n, dtype=H16, device=p0.device)
    if (src.data_ptr() & 15) == 0:
        L.call("vpf_cast_f32_h16", src, out, n)
    else:   # unaligned view: stage through an aligned copy
        L.call("vpf_cast_f32_h16", src.clone(), out, n)
    p0._vpf_shadow = (ver, out)
    return out


# --------------------------------------------------------------------------- thin kernel wrappers
def to_h16(x: torch.Tensor) -> torch.Tensor:
    if x.dtype == H16:
        return x.contiguous()
    x = x.contiguous().float()
    out = torch.empty(x.shape, dtype=H16, device=x.device)
    if x.numel():
        L.call("vpf_cast_f32_h16", x, out, x.numel())
    return out


def to_f32(x: torch.Tensor) -> torch.Tensor:
    if x.dtype == F32:
        return x.contiguous()
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=F32, device=x.device)
    if x.numel():
        L.call("vpf_cast_h16_f32", x, out, x.numel())
    return out


def gemm(A, a_tr, lda, Bm, b_tr, ldb, M, N, K, C, ldc, *, c_f32, mode=EPI_STORE, bias=None, C2=None, ldc2=0, res=None,
         ldres=0, aux=None, ldaux=0, gbias=None, group=1, site=0, p=0.0, splitk=0, batch=1, sAb=0, sBb=0, sCb=0, dbias=None,
         rng_state=None):
    st = (rng_state if rng_state is not None else rng.state(C.device)) if mode == EPI_DROP_RES else None
    L.call("vpf_gemm_h16", A, int(a_tr), lda, Bm, int(b_tr), ldb, M, N, K, batch, sAb, sBb, sCb, C, ldc, int(c_f32), mode,
           bias, C2, ldc2, res, ldres, aux, ldaux, gbias, group, st, site, float(p), splitk, dbias)


def gemm_fused(A, a_tr, lda, Bm, b_tr, ldb, M, N, K, C, ldc, *, c_f32, mode=EPI_STORE, a_kind=0, a_ab=None, a_dout=None, a_arg=None,
               a_group=1, a_ncols=0, b_kind=0, b_ab=None, bias=None, C2=None, ldc2=0, group=1, splitk=0, dbias=None):
    """vpf_gemm_h16_fused: operand prologues (BatchNorm+ReLU affine / virtual max-pool gradient) + group-max epilogue."""
    nch_a = a_ab.numel() // 2 if a_ab is not None else 0
    nch_b = b_ab.numel() // 2 if b_ab is not None else 0
    L.call("vpf_gemm_h16_fused", A, int(a_tr), lda, a_kind, a_ab, a_ab[nch_a:] if a_ab is not None else None, a_dout, a_arg,
           a_group, a_ncols, Bm, int(b_tr), ldb, b_kind, b_ab, b_ab[nch_b:] if b_ab is not None else None, M, N, K, C, ldc,
           int(c_f32), mode, bias, C2, ldc2, group, splitk, dbias)


EPI_GROUPMAX = 7


def linear_fwd(x16, w16, N, K, bias=None, *, out_f32=False, mode=EPI_STORE, **kw):
    """y[M,N] = x16[M,K] @ w16[N,K]^T (+bias, epilogue)."""
    M = x16.numel() // K
    y = torch.empty(M, N, dtype=F32 if out_f32 else H16, device=x16.device)
    gemm(x16, 0, K, w16, 0, K, M, N, K, y, N, c_f32=out_f32, mode=mode, bias=bias, **kw)
    return y


def linear_dgrad(dy16, w16, N, K, *, out_f32=False, mode=EPI_STORE, **kw):
    """dx[M,K] = dy16[M,N] @ w16[N,K]   (W read k-strided: no transposed copy)."""
    M = dy16.numel() // N
    dx = torch.empty(M, K, dtype=F32 if out_f32 else H16, device=dy16.device)
    gemm(dy16, 0, N, w16, 1, K, M, K, N, dx, K, c_f32=out_f32, mode=mode, **kw)
    return dx


# Weight gradients are off the backward critical path (nothing downstream reads them before the optimizer), so they CAN be
# issued on a side stream per compute stream (dgrad chain on the main stream, wgrad GEMMs on idle CUs).  Optional:
_wgrad_streams = {}


def _wgrad_side():
    cur = torch.cuda.current_stream()
    side = _wgrad_streams.get(cur.cuda_stream)
    if side is None:
        side = torch.cuda.Stream(device=cur.device)
        _wgrad_streams[cur.cuda_stream] = side
    return cur, side


def join_wgrad_streams() -> None:
    """Make the current stream wait for every outstanding side-stream weight gradient (call before the optimizer)."""
    cur = torch.cuda.current_stream()
    for side in _wgrad_streams.values():
        cur.wait_stream(side)


def linear_wgrad(dy16, x16, N, K, dW, dbias=None):
    """dW[N,K] += dy16[M,N]^T @ x16[M,K]   (both operands k-strided, split over M, fp32 atomics);
    dbias[N] += column sums of dy16 in the same pass."""
    M = dy16.numel() // N
    if not cfg.wgrad_async:
        gemm(dy16, 1, N, x16, 1, K, N, K, M, dW, K, c_f32=True, mode=EPI_ATOMIC, dbias=dbias)
        return
    cur, side = _wgrad_side()
    side.wait_stream(cur)                                   # dy16 / x16 are produced on the compute stream
    with torch.cuda.stream(side):
        gemm(dy16, 1, N, x16, 1, K, N, K, M, dW, K, c_f32=True, mode=EPI_ATOMIC, dbias=dbias)
    dy16.record_stream(side)
    x16.record_stream(side)




WGRAD_WS_BYTES = 4096 + 1024 * 65536      # arrival counters + one 128 x 128 f32 partial tile for up to 1024 workgroups
_wgrad_ws = {}


def wgrad_workspace(device):
    """The split-K scratch of vpf_wgrad_group, private to (device, current stream): two branches that run their grouped weight
    gradients concurrently on two streams must not share partial tiles.  Counters zeroed once; the kernel leaves them zero.
    None unless cfg.wgrad_deterministic: the library then adds the slices' tiles into dW with fp32 atomics."""
    if not cfg.wgrad_deterministic:
        return None
    key = (_Rng._index(device), torch.cuda.current_stream().cuda_stream)
    ws = _wgrad_ws.get(key)
    if ws is None:
        ws = torch.empty(WGRAD_WS_BYTES // 4, dtype=F32, device=f"cuda:{key[0]}")
        ws[:1024].zero_()
        _wgrad_ws[key] = ws
    return ws


class WgradDeferral:
    """Weight gradients are off the backward critical path: nothing reads them before the optimizer.  A trainer that runs two
    branches on two streams opens a deferral around ``backward``: grouped weight-gradient launches issued from streams other than
    ``target`` are not launched where they arise but queued (with an event marking the point their operands are complete), and
    ``drain`` issues them on ``target`` -- the stream of the SHORTER branch, behind that branch's own work.  The long branch's
    stream then carries only its dgrad chain.  No extra stream is created (a third stream costs more than it overlaps on this
    stack)."""

    def __init__(self, target: torch.cuda.Stream):
        self.target = target
        self.queue = []

    def take(self, jobs, arr) -> bool:
        cur = torch.cuda.current_stream()
        if cur == self.target:
            return False
        ev = torch.cuda.Event()
        ev.record(cur)
        self.queue.append((ev, jobs, arr))
        return True

    def drain(self) -> None:
        with torch.cuda.stream(self.target):
            for ev, jobs, arr in self.queue:
                self.target.wait_event(ev)
                ws = wgrad_workspace(jobs[0][0].device)
                L.call_struct("vpf_wgrad_group", arr, len(jobs), ws.data_ptr() if ws is not None else None, ws.numel() * 4 if ws is not None else 0)
                for dy, x, *_ in jobs:
                    dy.record_stream(self.target)
                    x.record_stream(self.target)
        self.queue = []




class KvBwdDeferral:
    """AdapterKVFn.backward hands its arguments over instead of launching; ``drain`` runs it on ``target`` (the image branch's stream,
    behind that branch's backward) once ``backward`` has returned, not before ``gate`` (an event on the point-cloud stream, recorded by
    Group2Emb's backward behind its chip-filling kernels).  See AdapterKVFn.backward."""

    def __init__(self, target: torch.cuda.Stream):
        self.target = target
        self.queue = []
        self.gate = None
        self.ready = None

    def mark_ready(self) -> None:
        """The producer of the node's input gradient (the cross-attention backward) has been launched on the current stream: the
        deferred chain waits for THIS point, not for the point where autograd gets round to the node (the end of the whole backward
        pass when the node was created first)."""
        self.ready = torch.cuda.Event()
        self.ready.record(torch.cuda.current_stream())

    def take(self, args) -> bool:
        cur = torch.cuda.current_stream()
        if cur == self.target:
            return False
        ev = self.ready
        if ev is None:
            ev = torch.cuda.Event()
            ev.record(cur)
        self.ready = None
        self.queue.append((ev, args))
        return True

    def mark_gate(self) -> None:
        self.gate = torch.cuda.Event()
        self.gate.record(torch.cuda.current_stream())

    def drain(self) -> None:
        with torch.cuda.stream(self.target):
            for ev, args in self.queue:
                self.target.wait_event(ev)
                if self.gate is not None:
                    self.target.wait_event(self.gate)
                AdapterKVFn._backward(*args)
                for t in tuple(args[3]) + (args[4],):
                    t.record_stream(self.target)          # (allocated on the main stream, read by this stream's kernels)
        self.queue = []


class WgradBatch:
    """Collects linear_wgrad calls of one layer and issues them as ONE grouped launch (vpf_wgrad_group)."""

    CAP = 32                               # GEMM_GROUP_MAX of csrc/gemm.hip

    def __init__(self, cap: int = 8):
        self.jobs = []
        self.cap = min(cap, self.CAP)      # 8: one layer's launch; 32: a whole encoder stack's (cfg.wgrad_stack)

    def add(self, dy16, x16, N, K, dW, dbias=None):
        if not cfg.wgrad_group:
            linear_wgrad(dy16, x16, N, K, dW, dbias)
            return
        self.jobs.append((dy16, x16, dy16.numel() // N, N, K, dW, dbias))
        if len(self.jobs) == self.cap:
            self.flush()

    def flush(self):
        if not self.jobs:
            return
        arr = (L.WgradJob * self.CAP)()
        for i, (dy, x, M, N, K, dW, db) in enumerate(self.jobs):
            arr[i].dy, arr[i].x, arr[i].M, arr[i].N, arr[i].K = dy.data_ptr(), x.data_ptr(), M, N, K
            arr[i].dW, arr[i].dbias = dW.data_ptr(), (db.data_ptr() if db is not None else None)
        if cfg.wgrad_defer is not None and cfg.wgrad_defer.take(self.jobs, arr):
            self.jobs = []
            return
        if cfg.wgrad_async or cfg.wgrad_group_async:
            # weight gradients are off the dgrad critical path: one grouped launch per layer on a side stream
            cur, side = _wgrad_side()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                ws = wgrad_workspace(self.jobs[0][0].device)
                L.call_struct("vpf_wgrad_group", arr, len(self.jobs), ws.data_ptr() if ws is not None else None, ws.numel() * 4 if ws is not None else 0)
            for dy, x, *_ in self.jobs:
                dy.record_stream(side)
                x.record_stream(side)
        else:
            ws = wgrad_workspace(self.jobs[0][0].device)
            L.call_struct("vpf_wgrad_group", arr, len(self.jobs), ws.data_ptr() if ws is not None else None, ws.numel() * 4 if ws is not None else 0)
        self.jobs = []


def colsum(x, C, acc, acc2=None):
    M = x.numel() // C
    L.call("vpf_colsum", x, int(x.dtype == H16), M, C, acc, acc2)


def layernorm_fwd(x, gamma, beta, pos=None, want_sum=False):
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty(x.shape, dtype=H16, device=x.device)
    mean = torch.empty(rows, dtype=F32, device=x.device)
    rstd = torch.empty(rows, dtype=F32, device=x.device)
    xsum = torch.empty(x.shape, dtype=F32, device=x.device) if (pos is not None and want_sum) else None
    pos_rows = (pos.numel() // D) if pos is not None else 0
    L.call("vpf_layernorm_fwd", x, int(x.dtype == H16), pos, pos_rows, gamma, beta, y, xsum, mean, rstd, rows, D, 1e-5)
    return y, mean, rstd, xsum


def layernorm_bwd(dy16, x, mean, rstd, gamma_p, beta_p, dres=None, out_h16=False):
    D = x.shape[-1]
    rows = x.numel() // D
    dx = torch.empty(x.shape, dtype=H16 if out_h16 else F32, device=x.device)
    ws = torch.empty(2 * 1024 * D, dtype=F32, device=x.device)      # per-block dgamma/dbeta partials
    L.call("vpf_layernorm_bwd", dy16, x, int(x.dtype == H16), mean, rstd, gamma_p.data, dres, dx, int(out_h16),
           grad_buf(gamma_p), grad_buf(beta_p), ws, ws.numel(), rows, D)
    return dx


# --------------------------------------------------------------------------- attention residual block
class AttnBlockFn(torch.autograd.Function):
    """[pos-add] -> LN(q) [, LN(kv)] -> q/k/v projections -> fused attention -> o_proj [-> dropout + residual].

    Covers MultiHeadAttention, CrossAttention, SelfAttention and Residual(attention) of
    partseg.py:14-141,201-213 depending on the flags in ``cfg``."""

    @staticmethod
    def forward(ctx, xq, pos, xkv, mod, cfg, *params):
        ctx.nparams, ctx.params = len(params), params
        # mod: the MultiHeadAttention module (weights); cfg: dict(ln_q, ln_kv (modules or None), residual, p_res, site_res, training)
        B, Lq, D = xq.shape
        H = mod.num_heads
        dev = xq.device
        training = cfg["training"]
        st = ctx.rng_st = rng.acquire(dev, training)
        is_self = xkv is None
        lnq, lnkv = cfg["ln_q"], cfg["ln_kv"]
        xq = xq.contiguous()
        base = xq
        mq = rq = None
        if lnq is not None:
            nq, mq, rq, xsum = layernorm_fwd(xq, lnq.weight.data, lnq.bias.data, pos=pos, want_sum=True)
            if xsum is not None:
                base = xsum
        else:
            if pos is not None:
                raise L.VpfError("pos requires a query LayerNorm")
            nq = to_h16(xq)
        mk = rk = None
        if is_self:
            nk, Lkv = nq, Lq
        else:
            xkv = xkv.contiguous()
            Lkv = xkv.shape[1]
            if lnkv is not None:
                nk, mk, rk, _ = layernorm_fwd(xkv, lnkv.weight.data, lnkv.bias.data)
            else:
                nk = to_h16(xkv)
        qkvw = [mod.q_proj.weight, mod.k_proj.weight, mod.v_proj.weight]
        pack_params(qkvw)
        w16 = shadow(qkvw)                      # [3D, D] h16
        Mq, Mk = B * Lq, B * Lkv
        if is_self:
            qkv = linear_fwd(nq, w16, 3 * D, D)                                  # [Mq, 3D]
            q, k, v, ldq, ldk, ldv = qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, 3 * D, 3 * D
            kv = None
        else:
            qkv = linear_fwd(nq, w16[:D * D], D, D)                              # [Mq, D]
            kv = linear_fwd(nk, w16[D * D:], 2 * D, D)                           # [Mk, 2D]
            q, k, v, ldq, ldk, ldv = qkv, kv, kv[:, D:], D, 2 * D, 2 * D
        o = torch.empty(Mq, D, dtype=H16, device=dev)
        lse = torch.empty(B * H * Lq, dtype=F32, device=dev)
        p_att = mod.dropout.p if training else 0.0
        pad = ctx.pad = cfg.get("pad_mask")
        if pad is None:
            L.call("vpf_attention_fwd", q, ldq, k, ldk, v, ldv, B, H, Lq, Lkv, D // H, float(mod.dp_scale), float(p_att),
                   st, mod.site_attn, o, D, lse)
        else:
            L.call("vpf_attention_fwd_pad", q, ldq, k, ldk, v, ldv, B, H, Lq, Lkv, D // H, float(mod.dp_scale), float(p_att),
                   st, mod.site_attn, o, D, lse, pad)
        wo16 = shadow([mod.o_proj.weight])
        residual = cfg["residual"]
        p_res = cfg["p_res"] if training else 0.0
        if residual:
            out = linear_fwd(o, wo16, D, D, mod.o_proj.bias.data, out_f32=True, mode=EPI_DROP_RES, res=base, ldres=D,
                             site=cfg["site_res"], p=p_res, rng_state=st)
        else:
            out = linear_fwd(o, wo16, D, D, mod.o_proj.bias.data, out_f32=True)
        ctx.mod, ctx.cfg = mod, cfg
        ctx.dims = (B, Lq, Lkv, D, H, is_self, p_att, p_res)
        ctx.has_pos = pos is not None
        ctx.pos_shape = tuple(pos.shape) if pos is not None else None
        ctx.xkv_dtype = None if is_self else xkv.dtype
        ctx.save_for_backward(base, mq, rq, nq, None if is_self else xkv, mk, rk, None if is_self else nk, qkv, kv, o, lse)
        return out.view(B, Lq, D)

    @staticmethod
    @_sinked
    def backward(ctx, dout):
        base, mq, rq, nq, xkv, mk, rk, nk, qkv, kv, o, lse = ctx.saved_tensors
        mod, cfg = ctx.mod, ctx.cfg
        B, Lq, Lkv, D, H, is_self, p_att, p_res = ctx.dims
        dev = dout.device
        dout = dout.contiguous().float()
        Mq, Mk = B * Lq, B * Lkv
        residual = cfg["residual"]
        if residual and p_res > 0.0:
            dz = torch.empty(Mq, D, dtype=H16, device=dev)
            L.call("vpf_dropout_bwd", dout, dz, dout.numel(), ctx.rng_st, cfg["site_res"], float(p_res))
        else:
            dz = to_h16(dout).view(Mq, D)
        wg = WgradBatch()                      # the block's weight gradients as ONE grouped launch at the end (they are off the dgrad chain)
        wg.add(dz, o, D, D, grad_buf(mod.o_proj.weight), grad_buf(mod.o_proj.bias))
        do = linear_dgrad(dz, shadow([mod.o_proj.weight]), D, D)
        qkvw = [mod.q_proj.weight, mod.k_proj.weight, mod.v_proj.weight]
        w16 = shadow(qkvw)
        gW = packed_grad(qkvw)
        if is_self:
            dqkv = torch.empty(Mq, 3 * D, dtype=H16, device=dev)
            q, k, v, ldq, ldk, ldv = qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, 3 * D, 3 * D
            dq, dk, dv, lddq, lddk, lddv = dqkv, dqkv[:, D:], dqkv[:, 2 * D:], 3 * D, 3 * D, 3 * D
        else:
            dq = torch.empty(Mq, D, dtype=H16, device=dev)
            dkv = torch.empty(Mk, 2 * D, dtype=H16, device=dev)
            q, k, v, ldq, ldk, ldv = qkv, kv, kv[:, D:], D, 2 * D, 2 * D
            dk, dv, lddq, lddk, lddv = dkv, dkv[:, D:], D, 2 * D, 2 * D
        if ctx.pad is None:
            L.call("vpf_attention_bwd", q, ldq, k, ldk, v, ldv, o, D, do, D, lse, B, H, Lq, Lkv, D // H, float(mod.dp_scale),
                   float(p_att), ctx.rng_st, mod.site_attn, dq, lddq, dk, lddk, dv, lddv,
                   torch.empty(B * H * Lq, dtype=F32, device=dev))
        else:
            L.call("vpf_attention_bwd_pad", q, ldq, k, ldk, v, ldv, o, D, do, D, lse, B, H, Lq, Lkv, D // H, float(mod.dp_scale),
                   float(p_att), ctx.rng_st, mod.site_attn, dq, lddq, dk, lddk, dv, lddv,
                   torch.empty(B * H * Lq, dtype=F32, device=dev), ctx.pad)
        dxkv = None
        if is_self:
            wg.add(dqkv, nq, 3 * D, D, gW)
            dnq = linear_dgrad(dqkv, w16, 3 * D, D)
        else:
            wg.add(dq, nq, D, D, gW[:D * D])
            dnq = linear_dgrad(dq, w16[:D * D], D, D)
            wg.add(dkv, nk, 2 * D, D, gW[D * D:])
            if ctx.needs_input_grad[2]:
                dnk = linear_dgrad(dkv, w16[D * D:], 2 * D, D)
                lnkv = cfg["ln_kv"]
                if lnkv is not None:
                    dxkv = layernorm_bwd(dnk, xkv, mk, rk, lnkv.weight, lnkv.bias, None, out_h16=(ctx.xkv_dtype == H16))
                    dxkv = dxkv.view(B, Lkv, D)
                else:
                    dxkv = (dnk if ctx.xkv_dtype == H16 else to_f32(dnk)).view(B, Lkv, D)
            elif cfg["ln_kv"] is not None:
                # kv input needs no grad but the kv LayerNorm affine does
                dnk = linear_dgrad(dkv, w16[D * D:], 2 * D, D)
                lnkv = cfg["ln_kv"]
                layernorm_bwd(dnk, xkv, mk, rk, lnkv.weight, lnkv.bias, None, out_h16=True)
        wg.flush()
        lnq = cfg["ln_q"]
        if lnq is not None:
            dbase = layernorm_bwd(dnq, base, mq, rq, lnq.weight, lnq.bias, dout if residual else None)
        else:
            dbase = to_f32(dnq)
            if residual:
                dbase = dbase.view_as(dout) + dout
        dbase = dbase.view(B, Lq, D)
        dpos = None
        if ctx.has_pos and ctx.needs_input_grad[1]:
            if ctx.pos_shape[0] == B or B == 1:
                dpos = dbase.view(ctx.pos_shape)
            else:
                dpos = torch.zeros(ctx.pos_shape, dtype=F32, device=dev)
                L.call("vpf_rowsum_mod_f32", dbase, Mq, D, Lq, dpos)
        return (dbase, dpos, dxkv, None, None) + (None,) * ctx.nparams


# --------------------------------------------------------------------------- MLP residual block
class MLPBlockFn(torch.autograd.Function):
    """LN -> Linear -> GELU -> Linear [-> dropout + residual]   (partseg.py:191-213)."""

    @staticmethod
    def forward(ctx, x, mod, cfg, *params):
        ctx.nparams, ctx.params = len(params), params
        shp = x.shape
        D = shp[-1]
        x = x.contiguous().float()
        ln, fc1, fc2 = mod[0], mod[1], mod[3]
        Hd = fc1.weight.shape[0]
        n, mean, rstd, _ = layernorm_fwd(x, ln.weight.data, ln.bias.data)
        M = x.numel() // D
        u = torch.empty(M, Hd, dtype=H16, device=x.device)
        h = linear_fwd(n, shadow([fc1.weight]), Hd, D, fc1.bias.data, mode=EPI_GELU, C2=u, ldc2=Hd)
        training, residual = cfg["training"], cfg["residual"]
        p_res = cfg["p_res"] if training else 0.0
        ctx.rng_st = rng.acquire(x.device, training and residual and p_res > 0.0)
        if residual:
            out = linear_fwd(h, shadow([fc2.weight]), D, Hd, fc2.bias.data, out_f32=True, mode=EPI_DROP_RES, res=x, ldres=D,
                             site=cfg["site_res"], p=p_res, rng_state=ctx.rng_st)
        else:
            out = linear_fwd(h, shadow([fc2.weight]), D, Hd, fc2.bias.data, out_f32=True)
        ctx.mod, ctx.cfg, ctx.p_res = mod, cfg, p_res
        ctx.save_for_backward(x, mean, rstd, n, u, h)
        return out.view(shp)

    @staticmethod
    @_sinked
    def backward(ctx, dout):
        x, mean, rstd, n, u, h = ctx.saved_tensors
        mod, cfg, p_res = ctx.mod, ctx.cfg, ctx.p_res
        ln, fc1, fc2 = mod[0], mod[1], mod[3]
        D = x.shape[-1]
        Hd = fc1.weight.shape[0]
        M = x.numel() // D
        dout = dout.contiguous().float()
        residual = cfg["residual"]
        if residual and p_res > 0.0:
            dz = torch.empty(M, D, dtype=H16, device=x.device)
            L.call("vpf_dropout_bwd", dout, dz, dout.numel(), ctx.rng_st, cfg["site_res"], float(p_res))
        else:
            dz = to_h16(dout).view(M, D)
        wg = WgradBatch()
        wg.add(dz, h, D, Hd, grad_buf(fc2.weight), grad_buf(fc2.bias))
        du = linear_dgrad(dz, shadow([fc2.weight]), D, Hd, mode=EPI_GELU_BWD, aux=u, ldaux=Hd)
        wg.add(du, n, Hd, D, grad_buf(fc1.weight), grad_buf(fc1.bias))
        dn = linear_dgrad(du, shadow([fc1.weight]), Hd, D)
        wg.flush()
        dx = layernorm_bwd(dn, x, mean, rstd, ln.weight, ln.bias, dout if residual else None)
        return (dx.view_as(dout), None, None) + (None,) * ctx.nparams


# --------------------------------------------------------------------------- fused self-attention stack


def sa_stack_supported(layers, x) -> bool:
    """The row-block kernels cover D = 256 / 4 heads / hidden 512 and D = 384 / 6 heads / hidden 1536 (heads of 64: every BASELINE
    architecture), sequences of <= 224 tokens (the resident attention kernels), identity drop-path."""
    if not cfg.sa_fused or len(layers) == 0 or x.dim() != 3:
        return False
    B, Lq, D = x.shape
    if D not in FUSED_WIDTHS or Lq > 224:
        return False
    first = None
    for layer in layers:
        att = layer[0].module.attention
        mlp = layer[1].module
        hw = (att.num_heads, mlp[1].weight.shape[0])
        if hw not in FUSED_WIDTHS[D] or mlp[1].weight.shape[1] != D or hw != (first or hw):
            return False
        first = hw
        if not isinstance(layer[0].drop_path, torch.nn.Identity) or not isinstance(layer[1].drop_path, torch.nn.Identity):
            return False
    return True


# model width -> the (heads, MLP hidden) pairs the fused encoder kernels are built for (first = the default of that width; D = 256 with
# hidden 1024 = mlp_widen_factor 4, scripts/pretrain/pt-*-MR4-0.sh of the reference, runs the D-generic kernels of sa_rows.hip)
FUSED_WIDTHS = {256: ((4, 512), (4, 1024)), 384: ((6, 1536),)}


def _block_dims(att, mlp):
    """(D, hidden, heads) of one attention + MLP block."""
    return mlp[1].weight.shape[1], mlp[1].weight.shape[0], att.num_heads


def pgrad_rows(M: int, D: int, hidden: Optional[int] = None) -> int:
    """Partial rows the backward row-block kernels write per LayerNorm (vpf_sa_layer_pgrad_rows[_h])."""
    if hidden is None:
        hidden = FUSED_WIDTHS[D][0][1] if D in FUSED_WIDTHS else 2 * D
    fn = L.lib().vpf_sa_layer_pgrad_rows_h
    fn.argtypes, fn.restype = [ctypes.c_long, ctypes.c_int, ctypes.c_int], ctypes.c_int
    return int(fn(M, D, hidden))


def _sa_packed(layers, dev):
    """Fragment-order copies of the stack's weights -- forward: Wo, W1, W2 (and Wqkv of layers >= 1); backward (dgrad):
    the transposed views W2T, W1T, WoT, WqkvT -- rewritten from the h16 shadow on every call (the shadow changes every
    optimizer step; the copy is one small kernel)."""
    holder = layers[0]
    D, Hd, _ = _block_dims(layers[0][0].module.attention, layers[0][1].module)
    sizes = [("Wo", D * D), ("W1", Hd * D), ("W2", D * Hd), ("Wqkv", 3 * D * D), ("W2T", Hd * D), ("W1T", D * Hd), ("WoT", D * D), ("WqkvT", 3 * D * D)]
    per_layer = sum(n for _, n in sizes)
    buf = getattr(holder, "_vpf_packed", None)
    if buf is None or buf.device != dev or buf.numel() != per_layer * len(layers):
        buf = torch.empty(per_layer * len(layers), dtype=H16, device=dev)
        holder._vpf_packed = buf
    jobs = (L.PackJob * 64)()
    views = []
    n = 0
    for i, layer in enumerate(layers):
        att, mlp = layer[0].module.attention, layer[1].module
        qkvw = [att.q_proj.weight, att.k_proj.weight, att.v_proj.weight]
        pack_params(qkvw)
        o = i * per_layer
        v = {}
        for name, cnt in sizes:
            v[name] = buf[o:o + cnt]
            o += cnt
        views.append(v)
        wo, w1, w2, wqkv = shadow([att.o_proj.weight]), shadow([mlp[1].weight]), shadow([mlp[3].weight]), shadow(qkvw)
        #       src   dst        N (rows of the logical operand)  K (contraction)  transposed
        todo = [(wo, v["Wo"], D, D, 0), (w1, v["W1"], Hd, D, 0), (w2, v["W2"], D, Hd, 0),
                (w2, v["W2T"], Hd, D, 1), (w1, v["W1T"], D, Hd, 1), (wo, v["WoT"], D, D, 1), (wqkv, v["WqkvT"], D, 3 * D, 1)]
        if i > 0:
            todo.append((wqkv, v["Wqkv"], 3 * D, D, 0))
        for src, dst, N, K, tr in todo:
            jobs[n].src, jobs[n].dst, jobs[n].N, jobs[n].K, jobs[n].transposed = src.data_ptr(), dst.data_ptr(), N, K, tr
            n += 1
            if n == 64:
                L.call_struct("vpf_pack_wfrag", jobs, n)
                n = 0
    if n:
        L.call_struct("vpf_pack_wfrag", jobs, n)
    return views


class SAStackFn(torch.autograd.Function):
    """The self-attention stack of Encoder.forward (partseg.py:326-340: pos re-added before every SelfAttentionLayer,
    :170-188) with ONE forward kernel per layer (vpf_sa_layer_fwd); backward runs the same dgrad / wgrad / attention /
    LayerNorm kernels as the unfused blocks on the tensors the fused forward saved."""

    @staticmethod
    def forward(ctx, x, pos, layers, training, *params):
        ctx.nparams, ctx.params = len(params), params
        B, Lq, D = x.shape
        M = B * Lq
        _, Hd, H = _block_dims(layers[0][0].module.attention, layers[0][1].module)
        dev = x.device
        x = x.contiguous().float()
        nl = len(layers)
        chunk_rows = Lq if Lq <= 96 else (Lq + 1) // 2
        packed = _sa_packed(layers, dev)
        st = ctx.rng_st = rng.acquire(dev, training)
        l0 = layers[0]
        ln1 = l0[0].module.norm
        att0 = l0[0].module.attention
        n1, m1, r1, xsum = layernorm_fwd(x, ln1.weight.data, ln1.bias.data, pos=pos, want_sum=True)
        base = xsum if xsum is not None else x
        qkvw0 = [att0.q_proj.weight, att0.k_proj.weight, att0.v_proj.weight]
        qkv = linear_fwd(n1, shadow(qkvw0), 3 * D, D)
        pos_c = pos.contiguous().float() if pos is not None else None
        pos_rows = pos_c.numel() // D if pos_c is not None else 0
        flat = []
        out = None
        for i, layer in enumerate(layers):
            att, mlp = layer[0].module.attention, layer[1].module
            ln2, fc1, fc2 = mlp[0], mlp[1], mlp[3]
            last = i + 1 == nl
            o = torch.empty(M, D, dtype=H16, device=dev)
            lse = torch.empty(B * H * Lq, dtype=F32, device=dev)
            x1 = torch.empty(M, D, dtype=F32, device=dev)
            m2 = torch.empty(M, dtype=F32, device=dev)
            r2 = torch.empty(M, dtype=F32, device=dev)
            n2 = torch.empty(M, D, dtype=H16, device=dev)
            u = torch.empty(M, Hd, dtype=H16, device=dev)
            h = torch.empty(M, Hd, dtype=H16, device=dev)
            out = torch.empty(M, D, dtype=F32, device=dev)
            a = L.SaLayerFwd()
            a.B, a.L, a.chunk_rows, a.D, a.H, a.hidden = B, Lq, chunk_rows, D, H, Hd
            a.qkv, a.base, a.rng = qkv.data_ptr(), base.data_ptr(), st.data_ptr()
            a.scale, a.p_att, a.site_att = float(att.dp_scale), float(att.dropout.p if training else 0.0), att.site_attn
            a.Wo, a.bo = packed[i]["Wo"].data_ptr(), att.o_proj.bias.data.data_ptr()
            a.p_res1, a.site_res1 = float(layer[0].dropout.p if training else 0.0), layer[0].site
            a.ln2_g, a.ln2_b = ln2.weight.data.data_ptr(), ln2.bias.data.data_ptr()
            a.W1, a.b1 = packed[i]["W1"].data_ptr(), fc1.bias.data.data_ptr()
            a.W2, a.b2 = packed[i]["W2"].data_ptr(), fc2.bias.data.data_ptr()
            a.p_res2, a.site_res2 = float(layer[1].dropout.p if training else 0.0), layer[1].site
            a.o, a.lse, a.x1, a.mean2, a.rstd2, a.n2 = o.data_ptr(), lse.data_ptr(), x1.data_ptr(), m2.data_ptr(), r2.data_ptr(), n2.data_ptr()
            a.u, a.h, a.out = u.data_ptr(), h.data_ptr(), out.data_ptr()
            split = cfg.sa_split_attn if (cfg.sa_split_attn is not None and D == 256 and Hd == 512) else True      # (attention inside the layer kernel: D = 256 / hidden 512 only)
            if split:
                L.call("vpf_attention_fwd", qkv, 3 * D, qkv[:, D:], 3 * D, qkv[:, 2 * D:], 3 * D, B, H, Lq, Lq, D // H, float(att.dp_scale),
                       float(a.p_att), st, att.site_attn, o, D, lse)
                a.attention_done = 1
            nxt = None
            if not last:
                lnn = layers[i + 1][0].module.norm
                nxt = (torch.empty(M, dtype=F32, device=dev), torch.empty(M, dtype=F32, device=dev),
                       torch.empty(M, D, dtype=H16, device=dev), torch.empty(M, 3 * D, dtype=H16, device=dev))
                a.pos, a.pos_rows = (pos_c.data_ptr() if pos_c is not None else None), pos_rows
                a.ln1n_g, a.ln1n_b, a.Wqkv_next = lnn.weight.data.data_ptr(), lnn.bias.data.data_ptr(), packed[i + 1]["Wqkv"].data_ptr()
                a.mean1n, a.rstd1n, a.n1n, a.qkv_next = nxt[0].data_ptr(), nxt[1].data_ptr(), nxt[2].data_ptr(), nxt[3].data_ptr()
            if cfg.sa_debug is not None:
                a.dbg = cfg.sa_debug.data_ptr()
            L.call_struct("vpf_sa_layer_fwd", a)
            flat += [base, m1, r1, n1, qkv, o, lse, x1, m2, r2, n2, u, h]
            if not last:
                base, m1, r1, n1, qkv = out, nxt[0], nxt[1], nxt[2], nxt[3]
        ctx.layers, ctx.training, ctx.packed = layers, training, packed
        ctx.dims = (B, Lq, D, Hd, H)
        ctx.pos_shape = tuple(pos.shape) if pos is not None else None
        ctx.save_for_backward(*flat)
        return out.view(B, Lq, D)

    @staticmethod
    @_sinked
    def backward(ctx, dout):
        flat = ctx.saved_tensors
        layers, training = ctx.layers, ctx.training
        B, Lq, D, Hd, H = ctx.dims
        M = B * Lq
        dev = dout.device
        d = dout.contiguous().float().view(M, D)
        st = ctx.rng_st
        want_pos = ctx.pos_shape is not None and ctx.needs_input_grad[1]
        if cfg.sa_fused_bwd:
            return SAStackFn._backward_fused(ctx, d, st, want_pos)
        dsum = None
        for i in range(len(layers) - 1, -1, -1):
            base, m1, r1, n1, qkv, o, lse, x1, m2, r2, n2, u, h = flat[13 * i:13 * i + 13]
            layer = layers[i]
            wg = WgradBatch()
            sa, mlp = layer[0].module, layer[1].module
            att, ln1 = sa.attention, sa.norm
            ln2, fc1, fc2 = mlp[0], mlp[1], mlp[3]
            # ---- MLP residual block (MLPBlockFn.backward)
            p2 = layer[1].dropout.p if training else 0.0
            if p2 > 0.0:
                dz = torch.empty(M, D, dtype=H16, device=dev)
                L.call("vpf_dropout_bwd", d, dz, d.numel(), st, layer[1].site, float(p2))
            else:
                dz = to_h16(d).view(M, D)
            wg.add(dz, h, D, Hd, grad_buf(fc2.weight), grad_buf(fc2.bias))
            du = linear_dgrad(dz, shadow([fc2.weight]), D, Hd, mode=EPI_GELU_BWD, aux=u, ldaux=Hd)
            wg.add(du, n2, Hd, D, grad_buf(fc1.weight), grad_buf(fc1.bias))
            dn = linear_dgrad(du, shadow([fc1.weight]), Hd, D)
            dx1 = layernorm_bwd(dn, x1, m2, r2, ln2.weight, ln2.bias, d)
            # ---- attention residual block (AttnBlockFn.backward, self-attention branch)
            p1 = layer[0].dropout.p if training else 0.0
            if p1 > 0.0:
                dz = torch.empty(M, D, dtype=H16, device=dev)
                L.call("vpf_dropout_bwd", dx1, dz, dx1.numel(), st, layer[0].site, float(p1))
            else:
                dz = to_h16(dx1).view(M, D)
            wg.add(dz, o, D, D, grad_buf(att.o_proj.weight), grad_buf(att.o_proj.bias))
            do = linear_dgrad(dz, shadow([att.o_proj.weight]), D, D)
            qkvw = [att.q_proj.weight, att.k_proj.weight, att.v_proj.weight]
            w16 = shadow(qkvw)
            dqkv = torch.empty(M, 3 * D, dtype=H16, device=dev)
            p_att = att.dropout.p if training else 0.0
            L.call("vpf_attention_bwd", qkv, 3 * D, qkv[:, D:], 3 * D, qkv[:, 2 * D:], 3 * D, o, D, do, D, lse, B, H, Lq, Lq, D // H,
                   float(att.dp_scale), float(p_att), st, att.site_attn, dqkv, 3 * D, dqkv[:, D:], 3 * D, dqkv[:, 2 * D:], 3 * D,
                   torch.empty(B * H * Lq, dtype=F32, device=dev))
            wg.add(dqkv, n1, 3 * D, D, packed_grad(qkvw))
            dn1 = linear_dgrad(dqkv, w16, 3 * D, D)
            d = layernorm_bwd(dn1, base, m1, r1, ln1.weight, ln1.bias, dx1).view(M, D)
            wg.flush()
            if want_pos:
                dsum = d.clone() if dsum is None else dsum.add_(d)
        dpos = None
        if want_pos:
            if ctx.pos_shape[0] == B or B == 1:
                dpos = dsum.view(ctx.pos_shape)
            else:
                dpos = torch.zeros(ctx.pos_shape, dtype=F32, device=dev)
                L.call("vpf_rowsum_mod_f32", dsum, M, D, Lq, dpos)
        return (d.view(B, Lq, D), dpos, None, None) + (None,) * ctx.nparams


    @staticmethod
    def _backward_fused(ctx, d, st, want_pos):
        flat = ctx.saved_tensors
        layers, training, packed = ctx.layers, ctx.training, ctx.packed
        B, Lq, D, Hd, H = ctx.dims
        M = B * Lq
        dev = d.device
        nwg = pgrad_rows(M, D, Hd)
        nl = len(layers)
        pg = torch.empty(nl, 2, nwg * 2 * D, dtype=F32, device=dev)       # LayerNorm parameter-gradient partials of every layer
        pjobs = (L.PgradJob * 32)()
        npj = 0
        dsum = torch.empty(M, D, dtype=F32, device=dev) if want_pos else None      # written (not accumulated) by the first layer processed
        for i in range(len(layers) - 1, -1, -1):
            base, m1, r1, n1, qkv, o, lse, x1, m2, r2, n2, u, h = flat[13 * i:13 * i + 13]
            layer = layers[i]
            sa, mlp = layer[0].module, layer[1].module
            att, ln1 = sa.attention, sa.norm
            ln2, fc1, fc2 = mlp[0], mlp[1], mlp[3]
            dz2 = torch.empty(M, D, dtype=H16, device=dev)
            du = torch.empty(M, Hd, dtype=H16, device=dev)
            dx1 = torch.empty(M, D, dtype=F32, device=dev)
            dz1 = torch.empty(M, D, dtype=H16, device=dev)
            do = torch.empty(M, D, dtype=H16, device=dev)
            dqkv = torch.empty(M, 3 * D, dtype=H16, device=dev)
            dbase = torch.empty(M, D, dtype=F32, device=dev)
            a = L.SaLayerBwd()
            a.M, a.D, a.hidden, a.rng = M, D, Hd, st.data_ptr()
            a.p_res1, a.site_res1 = float(layer[0].dropout.p if training else 0.0), layer[0].site
            a.p_res2, a.site_res2 = float(layer[1].dropout.p if training else 0.0), layer[1].site
            a.d, a.u, a.x1, a.mean2, a.rstd2, a.ln2_g = d.data_ptr(), u.data_ptr(), x1.data_ptr(), m2.data_ptr(), r2.data_ptr(), ln2.weight.data.data_ptr()
            a.W2T, a.W1T, a.WoT = packed[i]["W2T"].data_ptr(), packed[i]["W1T"].data_ptr(), packed[i]["WoT"].data_ptr()
            a.dz2, a.du, a.dx1, a.dz1, a.dout_attn = dz2.data_ptr(), du.data_ptr(), dx1.data_ptr(), dz1.data_ptr(), do.data_ptr()
            a.pgrad2, a.pgrad1 = pg[i, 1].data_ptr(), pg[i, 0].data_ptr()
            a.dqkv, a.WqkvT, a.base, a.mean1, a.rstd1, a.ln1_g = (dqkv.data_ptr(), packed[i]["WqkvT"].data_ptr(), base.data_ptr(), m1.data_ptr(),
                                                                 r1.data_ptr(), ln1.weight.data.data_ptr())
            a.dbase, a.dsum = dbase.data_ptr(), (dsum.data_ptr() if dsum is not None else None)
            a.dsum_init = int(i == len(layers) - 1)
            L.call_struct("vpf_sa_layer_bwd_mlp", a)
            p_att = att.dropout.p if training else 0.0
            L.call("vpf_attention_bwd", qkv, 3 * D, qkv[:, D:], 3 * D, qkv[:, 2 * D:], 3 * D, o, D, do, D, lse, B, H, Lq, Lq, D // H,
                   float(att.dp_scale), float(p_att), st, att.site_attn, dqkv, 3 * D, dqkv[:, D:], 3 * D, dqkv[:, 2 * D:], 3 * D,
                   torch.empty(B * H * Lq, dtype=F32, device=dev))
            L.call_struct("vpf_sa_layer_bwd_qkv", a)
            wg = WgradBatch()
            wg.add(dz2, h, D, Hd, grad_buf(fc2.weight), grad_buf(fc2.bias))
            wg.add(du, n2, Hd, D, grad_buf(fc1.weight), grad_buf(fc1.bias))
            wg.add(dz1, o, D, D, grad_buf(att.o_proj.weight), grad_buf(att.o_proj.bias))
            wg.add(dqkv, n1, 3 * D, D, packed_grad([att.q_proj.weight, att.k_proj.weight, att.v_proj.weight]))
            wg.flush()
            for which, ln in ((1, ln2), (0, ln1)):
                pjobs[npj].partials, pjobs[npj].rows, pjobs[npj].D = pg[i, which].data_ptr(), nwg, D
                pjobs[npj].dgamma, pjobs[npj].dbeta = grad_buf(ln.weight).data_ptr(), grad_buf(ln.bias).data_ptr()
                npj += 1
                if npj == 32:
                    L.call_struct("vpf_ln_pgrad_reduce", pjobs, npj)
                    npj = 0
            d = dbase
        if npj:
            L.call_struct("vpf_ln_pgrad_reduce", pjobs, npj)
        dpos = None
        if want_pos:
            if ctx.pos_shape[0] == B or B == 1:
                dpos = dsum.view(ctx.pos_shape)
            else:
                dpos = torch.zeros(ctx.pos_shape, dtype=F32, device=dev)
                L.call("vpf_rowsum_mod_f32", dsum, M, D, Lq, dpos)
        return (d.view(B, Lq, D), dpos, None, None) + (None,) * ctx.nparams


# --------------------------------------------------------------------------- fused encoder: cross-attention layer + self-attention stack
def _pack_blocks(blocks, holder, dev, front=None):
    """blocks: list of (attention module, MLP module, want_Wqkv_forward, want_WqkvT).  Returns per-block dicts of fragment-order
    weight views (see _sa_packed).  front = the position MLP's second Linear: its weight [D,128] is packed into views[0]["Wpos"] and
    block 0's q projection into views[0]["Wq"] (vpf_ca_front_fwd), in the same launch."""
    D, Hd, _ = _block_dims(blocks[0][0], blocks[0][1])
    sizes = [("Wo", D * D), ("W1", Hd * D), ("W2", D * Hd), ("Wqkv", 3 * D * D), ("W2T", Hd * D), ("W1T", D * Hd), ("WoT", D * D), ("WqkvT", 3 * D * D)]
    per = sum(n for _, n in sizes)
    extra = D * 128                                       # the position MLP's weight (always reserved: one buffer size per stack)
    buf = getattr(holder, "_vpf_packed_enc", None)
    if buf is None or buf.device != dev or buf.numel() != per * len(blocks) + extra:
        buf = torch.empty(per * len(blocks) + extra, dtype=H16, device=dev)
        holder._vpf_packed_enc = buf
    jobs = (L.PackJob * 64)()
    views, n = [], 0
    for i, (att, mlp, want_qkv, want_qkvT) in enumerate(blocks):
        qkvw = [att.q_proj.weight, att.k_proj.weight, att.v_proj.weight]
        pack_params(qkvw)
        o = i * per
        v = {}
        for name, cnt in sizes:
            v[name] = buf[o:o + cnt]
            o += cnt
        views.append(v)
        wo, w1, w2 = shadow([att.o_proj.weight]), shadow([mlp[1].weight]), shadow([mlp[3].weight])
        todo = [(wo, v["Wo"], D, D, 0), (w1, v["W1"], Hd, D, 0), (w2, v["W2"], D, Hd, 0),
                (w2, v["W2T"], Hd, D, 1), (w1, v["W1T"], D, Hd, 1), (wo, v["WoT"], D, D, 1)]
        if want_qkv:
            todo.append((shadow(qkvw), v["Wqkv"], 3 * D, D, 0))
        if want_qkvT:
            todo.append((shadow(qkvw), v["WqkvT"], D, 3 * D, 1))
        elif i == 0:
            # the cross-attention block: its WqkvT slot holds the transposed q projection alone (vpf_ca_front_bwd)
            v["WqT"] = v["WqkvT"][:D * D]
            todo.append((shadow(qkvw)[:D * D], v["WqT"], D, D, 1))
            v["WkvT"] = v["WqkvT"][D * D:]                  # ... and the transposed k | v projections (vpf_ca_kv_bwd)
            todo.append((shadow(qkvw)[D * D:], v["WkvT"], D, 2 * D, 1))
        for src, dst, N, K, tr in todo:
            jobs[n].src, jobs[n].dst, jobs[n].N, jobs[n].K, jobs[n].transposed = src.data_ptr(), dst.data_ptr(), N, K, tr
            n += 1
            if n == 64:
                L.call_struct("vpf_pack_wfrag", jobs, n)
                n = 0
    if front is not None:
        assert not blocks[0][2], "block 0's Wqkv slot holds the packed q projection"
        att0, v0 = blocks[0][0], views[0]
        v0["Wq"] = v0["Wqkv"][:D * D]                      # (the cross-attention block's own Wqkv slot is otherwise unused)
        v0["Wpos"] = buf[per * len(blocks):]
        wq = shadow([att0.q_proj.weight, att0.k_proj.weight, att0.v_proj.weight])[:D * D]
        for src, dst, N, K in ((wq, v0["Wq"], D, D), (shadow([front.weight]), v0["Wpos"], D, 128)):
            if n == 64:
                L.call_struct("vpf_pack_wfrag", jobs, n)
                n = 0
            jobs[n].src, jobs[n].dst, jobs[n].N, jobs[n].K, jobs[n].transposed = src.data_ptr(), dst.data_ptr(), N, K, 0
            n += 1
    if n:
        L.call_struct("vpf_pack_wfrag", jobs, n)
    return views


def _tail_fwd(att, mlp, res_attn, res_mlp, pk, training, st, B, Lq, qkv_dummy, base, o, lse, nxt, pos_c, pos_rows, dev):
    """One vpf_sa_layer_fwd launch with attention_done = 1.  nxt = (LayerNorm module, packed Wqkv) of the following
    self-attention layer or None.  Returns (saved tensors dict, out, next-head tuple or None)."""
    D, Hd, H = _block_dims(att, mlp)
    M = B * Lq
    ln2, fc1, fc2 = mlp[0], mlp[1], mlp[3]
    x1 = torch.empty(M, D, dtype=F32, device=dev)
    m2 = torch.empty(M, dtype=F32, device=dev)
    r2 = torch.empty(M, dtype=F32, device=dev)
    n2 = torch.empty(M, D, dtype=H16, device=dev)
    u = torch.empty(M, Hd, dtype=H16, device=dev)
    h = torch.empty(M, Hd, dtype=H16, device=dev)
    out = torch.empty(M, D, dtype=F32, device=dev)
    a = L.SaLayerFwd()
    a.B, a.L, a.chunk_rows, a.D, a.H, a.hidden = B, Lq, 64, D, H, Hd
    a.qkv, a.base, a.rng = qkv_dummy.data_ptr(), base.data_ptr(), st.data_ptr()
    a.scale, a.p_att, a.site_att = float(att.dp_scale), float(att.dropout.p if training else 0.0), att.site_attn
    a.Wo, a.bo = pk["Wo"].data_ptr(), att.o_proj.bias.data.data_ptr()
    a.p_res1, a.site_res1 = float(res_attn.dropout.p if training else 0.0), res_attn.site
    a.ln2_g, a.ln2_b = ln2.weight.data.data_ptr(), ln2.bias.data.data_ptr()
    a.W1, a.b1 = pk["W1"].data_ptr(), fc1.bias.data.data_ptr()
    a.W2, a.b2 = pk["W2"].data_ptr(), fc2.bias.data.data_ptr()
    a.p_res2, a.site_res2 = float(res_mlp.dropout.p if training else 0.0), res_mlp.site
    a.o, a.lse, a.x1, a.mean2, a.rstd2, a.n2 = o.data_ptr(), lse.data_ptr(), x1.data_ptr(), m2.data_ptr(), r2.data_ptr(), n2.data_ptr()
    a.u, a.h, a.out = u.data_ptr(), h.data_ptr(), out.data_ptr()
    a.attention_done = 1
    if cfg.sa_debug is not None:
        a.dbg = cfg.sa_debug.data_ptr()
    head = None
    if nxt is not None:
        lnn, wqkv = nxt
        head = (torch.empty(M, dtype=F32, device=dev), torch.empty(M, dtype=F32, device=dev),
                torch.empty(M, D, dtype=H16, device=dev), torch.empty(M, 3 * D, dtype=H16, device=dev))
        a.pos, a.pos_rows = (pos_c.data_ptr() if pos_c is not None else None), pos_rows
        a.ln1n_g, a.ln1n_b, a.Wqkv_next = lnn.weight.data.data_ptr(), lnn.bias.data.data_ptr(), wqkv.data_ptr()
        a.mean1n, a.rstd1n, a.n1n, a.qkv_next = head[0].data_ptr(), head[1].data_ptr(), head[2].data_ptr(), head[3].data_ptr()
    L.call_struct("vpf_sa_layer_fwd", a)
    return (x1, m2, r2, n2, u, h), out, head


                           # EncoderFusedFn.backward (i = -1: the cross-attention layer)


class EncoderFusedFn(torch.autograd.Function):
    """Encoder.forward (partseg.py:326-340) for one cross-attention layer + the self-attention stack: the cross-attention
    layer's tail (o_proj .. MLP) and every self-attention layer run as fused row-block kernels (vpf_sa_layer_fwd /
    vpf_sa_layer_bwd_*), each of which also prepares the next layer's LayerNorm + q/k/v projection; only the
    cross-attention front (two LayerNorms, q and kv projections) and the attention kernels themselves are separate."""

    @staticmethod
    def forward(ctx, x, pos, xkv, ca, layers, training, kv_ready, taps, *params):
        """kv_ready: ``xkv`` already holds the cross-attention K | V projections (h16 [B, Lkv, 2D], AdapterKVFn).
        taps: ascending 1-based self-attention layer numbers whose output is returned as well (Encoder.forward's ``layer_idx``,
        partseg.py:336-337).  Returns (x after the last layer, *tapped layer outputs)."""
        ctx.nparams, ctx.params = len(params), params
        ctx.kv_ready, ctx.taps = kv_ready, tuple(taps)
        ctx.set_materialize_grads(False)
        B, Lq, D = x.shape
        M = B * Lq
        dev = x.device
        x = x.contiguous().float()
        st = ctx.rng_st = rng.acquire(dev, training)
        nl = len(layers)
        cross, cmlp = ca[0].module, ca[1].module
        catt = cross.attention
        _, Hd, H = _block_dims(catt, cmlp)
        blocks = [(catt, cmlp, False, False)] + [(l[0].module.attention, l[1].module, True, True) for l in layers]
        pos_c = pos.contiguous().float() if pos is not None else None
        pos_rows = pos_c.numel() // D if pos_c is not None else 0
        stash = ca.__dict__.pop("_vpf_front_stash", None)            # one-shot: CaFrontFn left base / q_norm / q for THESE tensors
        if not (stash is not None and pos_c is not None and stash["tokens"] == x.data_ptr() and stash["pos"] == pos_c.data_ptr()
                and stash["shape"] == (B, Lq, D) and stash["nblocks"] == len(blocks)):
            stash = None
        packed = stash["packed"] if stash is not None else _pack_blocks(blocks, ca, dev)
        # ---- cross-attention front (AttnBlockFn.forward up to the attention)
        lnq, lnkv = cross.q_norm, cross.kv_norm
        qkvw = [catt.q_proj.weight, catt.k_proj.weight, catt.v_proj.weight]
        w16 = shadow(qkvw)
        if stash is not None:
            base_ca, mq, rq, nq, q = stash["base"], stash["mq"], stash["rq"], stash["nq"], stash["q"]
        else:
            nq, mq, rq, xsum = layernorm_fwd(x, lnq.weight.data, lnq.bias.data, pos=pos, want_sum=True)
            base_ca = xsum if xsum is not None else x
            q = linear_fwd(nq, w16[:D * D], D, D)
        xkv = xkv.contiguous()
        Lkv = xkv.shape[1]
        Mk = B * Lkv
        if kv_ready:
            kv = xkv.view(Mk, 2 * D)
            nk = mk = rk = xkv = torch.empty(0, device=dev)         # the kv side is AdapterKVFn's business
        else:
            nk, mk, rk, _ = layernorm_fwd(xkv, lnkv.weight.data, lnkv.bias.data)
            kv = linear_fwd(nk, w16[D * D:], 2 * D, D)
        o = torch.empty(M, D, dtype=H16, device=dev)
        lse = torch.empty(B * H * Lq, dtype=F32, device=dev)
        L.call("vpf_attention_fwd", q, D, kv, 2 * D, kv[:, D:], 2 * D, B, H, Lq, Lkv, D // H, float(catt.dp_scale),
               float(catt.dropout.p if training else 0.0), st, catt.site_attn, o, D, lse)
        nxt = (layers[0][0].module.norm, packed[1]["Wqkv"]) if nl else None
        saved_t, out, head = _tail_fwd(catt, cmlp, ca[0], ca[1], packed[0], training, st, B, Lq, q, base_ca, o, lse, nxt, pos_c, pos_rows, dev)
        flat = [base_ca, mq, rq, nq, xkv, mk, rk, nk, q, kv, o, lse] + list(saved_t)        # 18 tensors
        tapped = []
        # ---- self-attention layers
        for i, layer in enumerate(layers):
            att, mlp = layer[0].module.attention, layer[1].module
            base, (m1, r1, n1, qkv) = out, head
            o = torch.empty(M, D, dtype=H16, device=dev)
            lse = torch.empty(B * H * Lq, dtype=F32, device=dev)
            L.call("vpf_attention_fwd", qkv, 3 * D, qkv[:, D:], 3 * D, qkv[:, 2 * D:], 3 * D, B, H, Lq, Lq, D // H, float(att.dp_scale),
                   float(att.dropout.p if training else 0.0), st, att.site_attn, o, D, lse)
            nxt = (layers[i + 1][0].module.norm, packed[i + 2]["Wqkv"]) if i + 1 < nl else None
            saved_t, out, head = _tail_fwd(att, mlp, layer[0], layer[1], packed[i + 1], training, st, B, Lq, qkv, base, o, lse, nxt, pos_c,
                                           pos_rows, dev)
            flat += [base, m1, r1, n1, qkv, o, lse] + list(saved_t)                          # 13 per layer
            if i + 1 in ctx.taps:
                # the kernel stores x + pos for every layer but the last (the next layer's residual base, partseg.py:335); the tap is x
                tapped.append(out.view(B, Lq, D) - pos_c.view(-1, Lq, D) if (i + 1 < nl and pos_c is not None) else out.view(B, Lq, D).clone())
        ctx.ca, ctx.layers, ctx.training, ctx.packed = ca, layers, training, packed
        ctx.dims = (B, Lq, Lkv, D, Hd, H)
        ctx.pos_shape = tuple(pos.shape) if pos is not None else None
        ctx.xkv_dtype = H16 if kv_ready else xkv.dtype
        ctx.save_for_backward(*flat)
        return (out.view(B, Lq, D),) + tuple(tapped)

    @staticmethod
    @_sinked
    def backward(ctx, dout, *dtaps):
        flat = ctx.saved_tensors
        ca, layers, training, packed = ctx.ca, ctx.layers, ctx.training, ctx.packed
        B, Lq, Lkv, D, Hd, H = ctx.dims
        M, Mk = B * Lq, B * Lkv
        dev = flat[0].device
        st = ctx.rng_st
        tap_grad = {t: g for t, g in zip([t for t in ctx.taps if 1 <= t <= len(layers)], dtaps) if g is not None}
        d = dout.contiguous().float().view(M, D) if dout is not None else None
        want_pos = ctx.pos_shape is not None and ctx.needs_input_grad[1]
        nl = len(layers)
        nwg = pgrad_rows(M, D, Hd)
        pg = torch.empty(nl + 1, 2, nwg * 2 * D, dtype=F32, device=dev)
        pjobs = (L.PgradJob * 32)()
        npj = 0
        dsum = torch.empty(M, D, dtype=F32, device=dev) if want_pos else None      # written (not accumulated) by the first layer processed
        dsum_started = False
        # the stack's weight gradients: one grouped launch per layer, or (wgrad_stack) ONE for the whole stack at the end -- 7 x 32
        # output tiles fill the chip with 2 K slices each instead of 16, so the split-K flush is paid once (the operands of every
        # layer stay alive until then: ~75 MB per layer at c2)
        wg_all = WgradBatch(cap=WgradBatch.CAP) if (cfg.wgrad_stack and cfg.wgrad_group) else None      # (more than 32 jobs: a launch per 32)

        pending = [None]      # the SaLayerBwd of the layer above whose qkv half is still to be launched: it goes out together with the
                              # MLP half of the layer below (vpf_sa_layer_bwd_qkv_mlp: one launch per layer boundary)

        def bwd_mlp(a, blk, res_attn, res_mlp, pk, d, u, x1, m2, r2, slot):
            att, mlp = blk
            bufs = (torch.empty(M, D, dtype=H16, device=dev), torch.empty(M, Hd, dtype=H16, device=dev), torch.empty(M, D, dtype=F32, device=dev),
                    torch.empty(M, D, dtype=H16, device=dev), torch.empty(M, D, dtype=H16, device=dev))
            dz2, du, dx1, dz1, do = bufs
            a.M, a.D, a.hidden, a.rng = M, D, Hd, st.data_ptr()
            a.p_res1, a.site_res1 = float(res_attn.dropout.p if training else 0.0), res_attn.site
            a.p_res2, a.site_res2 = float(res_mlp.dropout.p if training else 0.0), res_mlp.site
            a.d, a.u, a.x1, a.mean2, a.rstd2, a.ln2_g = d.data_ptr(), u.data_ptr(), x1.data_ptr(), m2.data_ptr(), r2.data_ptr(), mlp[0].weight.data.data_ptr()
            a.W2T, a.W1T, a.WoT = pk["W2T"].data_ptr(), pk["W1T"].data_ptr(), pk["WoT"].data_ptr()
            a.dz2, a.du, a.dx1, a.dz1, a.dout_attn = dz2.data_ptr(), du.data_ptr(), dx1.data_ptr(), dz1.data_ptr(), do.data_ptr()
            a.pgrad2 = pg[slot, 1].data_ptr()
            above, pending[0] = pending[0], None
            if above is not None:
                L.call_struct("vpf_sa_layer_bwd_qkv_mlp", above[0], ctypes.addressof(a))
                pgrad_job(*above[1])                        # its partial rows exist only now (queued at defer time, a mid-loop
            else:                                           # flush of >= 32 jobs would have reduced unwritten memory)
                L.call_struct("vpf_sa_layer_bwd_mlp", a)
            return bufs

        def pgrad_job(slot, which, ln, partials=None, rows=None):
            """Queue the fold of one LayerNorm's parameter-gradient partial rows; ONLY behind the launch that writes them (a flush
            happens as soon as ``cfg.pgrad_flush`` jobs are queued)."""
            nonlocal npj
            pjobs[npj].partials, pjobs[npj].rows, pjobs[npj].D = (pg[slot, which] if partials is None else partials).data_ptr(), (rows or nwg), D
            pjobs[npj].dgamma, pjobs[npj].dbeta = grad_buf(ln.weight).data_ptr(), grad_buf(ln.bias).data_ptr()
            npj += 1
            if npj >= min(32, max(1, int(cfg.pgrad_flush))):
                L.call_struct("vpf_ln_pgrad_reduce", pjobs, npj)
                npj = 0

        for i in range(nl - 1, -1, -1):
            if cfg.enc_bwd_hook is not None:
                cfg.enc_bwd_hook(ca, i)
            tg = tap_grad.get(i + 1)
            if tg is not None:                              # the tapped output feeds the head AND the next layer
                tg = tg.contiguous().float().view(M, D)
                d = tg if d is None else d + tg
            if d is None:
                continue                                    # nothing downstream of this layer was used (untapped tail of the stack)
            base, m1, r1, n1, qkv, o, lse, x1, m2, r2, n2, u, h = flat[18 + 13 * i:18 + 13 * i + 13]
            layer = layers[i]
            sa, mlp = layer[0].module, layer[1].module
            att, ln1 = sa.attention, sa.norm
            pk = packed[i + 1]
            a = L.SaLayerBwd()
            dz2, du, dx1, dz1, do = bwd_mlp(a, (att, mlp), layer[0], layer[1], pk, d, u, x1, m2, r2, i + 1)
            dqkv = torch.empty(M, 3 * D, dtype=H16, device=dev)
            dbase = torch.empty(M, D, dtype=F32, device=dev)
            L.call("vpf_attention_bwd", qkv, 3 * D, qkv[:, D:], 3 * D, qkv[:, 2 * D:], 3 * D, o, D, do, D, lse, B, H, Lq, Lq, D // H,
                   float(att.dp_scale), float(att.dropout.p if training else 0.0), st, att.site_attn, dqkv, 3 * D, dqkv[:, D:], 3 * D,
                   dqkv[:, 2 * D:], 3 * D, torch.empty(B * H * Lq, dtype=F32, device=dev))
            a.dqkv, a.WqkvT, a.base, a.mean1, a.rstd1, a.ln1_g = (dqkv.data_ptr(), pk["WqkvT"].data_ptr(), base.data_ptr(), m1.data_ptr(),
                                                                 r1.data_ptr(), ln1.weight.data.data_ptr())
            a.dbase, a.dsum, a.pgrad1 = dbase.data_ptr(), (dsum.data_ptr() if dsum is not None else None), pg[i + 1, 0].data_ptr()
            a.dsum_init = int(not dsum_started)
            dsum_started = True
            if tap_grad.get(i) is None:
                pending[0] = (a, (i + 1, 0, ln1))           # launched with the MLP half of the layer below (its d is this dbase)
            else:
                L.call_struct("vpf_sa_layer_bwd_qkv", a)    # (a tapped boundary: the tap's gradient joins d first)
                pgrad_job(i + 1, 0, ln1)
            wg = wg_all if wg_all is not None else WgradBatch()
            wg.add(dz2, h, D, Hd, grad_buf(mlp[3].weight), grad_buf(mlp[3].bias))
            wg.add(du, n2, Hd, D, grad_buf(mlp[1].weight), grad_buf(mlp[1].bias))
            wg.add(dz1, o, D, D, grad_buf(att.o_proj.weight), grad_buf(att.o_proj.bias))
            wg.add(dqkv, n1, 3 * D, D, packed_grad([att.q_proj.weight, att.k_proj.weight, att.v_proj.weight]))
            if wg_all is None:
                wg.flush()
            pgrad_job(i + 1, 1, mlp[0])
            d = dbase
        # ---- cross-attention layer
        if cfg.enc_bwd_hook is not None:
            cfg.enc_bwd_hook(ca, -1)
        base_ca, mq, rq, nq, xkv, mk, rk, nk, q, kv, o, lse, x1, m2, r2, n2, u, h = flat[:18]
        cross, cmlp = ca[0].module, ca[1].module
        catt, lnq, lnkv = cross.attention, cross.q_norm, cross.kv_norm
        a = L.SaLayerBwd()
        dz2, du, dx1, dz1, do = bwd_mlp(a, (catt, cmlp), ca[0], ca[1], packed[0], d, u, x1, m2, r2, 0)
        pgrad_job(0, 1, cmlp[0])
        front_rows = (cfg.ca_front_bwd_fused and "WqT" in packed[0]                    # the query side as one row-block kernel below
                      and ((D == 256 and nwg == (M + 63) // 64) or (D == 384 and nwg == (M + 31) // 32)))      # (its partial rows: one per 64 / 32 tokens)
        if npj and not front_rows:
            L.call_struct("vpf_ln_pgrad_reduce", pjobs, npj)
            npj = 0
        dq = torch.empty(M, D, dtype=H16, device=dev)
        dkv = torch.empty(Mk, 2 * D, dtype=H16, device=dev)
        L.call("vpf_attention_bwd", q, D, kv, 2 * D, kv[:, D:], 2 * D, o, D, do, D, lse, B, H, Lq, Lkv, D // H, float(catt.dp_scale),
               float(catt.dropout.p if training else 0.0), st, catt.site_attn, dq, D, dkv, 2 * D, dkv[:, D:], 2 * D,
               torch.empty(B * H * Lq, dtype=F32, device=dev))
        if ctx.kv_ready and cfg.kv_bwd_defer is not None:
            cfg.kv_bwd_defer.mark_ready()          # dK | dV exist from here on (autograd may reach AdapterKVFn.backward much later)
        qkvw = [catt.q_proj.weight, catt.k_proj.weight, catt.v_proj.weight]
        w16, gW = shadow(qkvw), packed_grad(qkvw[:1] if ctx.kv_ready else qkvw)      # (K / V weights: AdapterKVFn's business then)
        wg = wg_all if wg_all is not None else WgradBatch()
        wg.add(dz2, h, D, Hd, grad_buf(cmlp[3].weight), grad_buf(cmlp[3].bias))
        wg.add(du, n2, Hd, D, grad_buf(cmlp[1].weight), grad_buf(cmlp[1].bias))
        wg.add(dz1, o, D, D, grad_buf(catt.o_proj.weight), grad_buf(catt.o_proj.bias))
        wg.add(dq, nq, D, D, gW[:D * D])
        if not ctx.kv_ready:
            wg.add(dkv, nk, 2 * D, D, gW[D * D:])                      # (image branch: K / V weights' gradient in the same grouped launch)
        if wg_all is None:
            wg.flush()
        dnq = None if front_rows else linear_dgrad(dq, w16[:D * D], D, D)
        dxkv = None
        if ctx.kv_ready:
            dxkv = dkv.view(B, Lkv, 2 * D)                              # AdapterKVFn.backward takes it from here
        elif front_rows and cfg.ca_kv_bwd_fused and "WkvT" in packed[0] and xkv.dtype == F32 and ctx.xkv_dtype == F32:
            # dk | dv . (Wk | Wv) -> kv LayerNorm' -> dxkv in one row-block kernel (vpf_ca_kv_bwd) instead of a GEMM and two LayerNorm launches
            dxkv = torch.empty(Mk, D, dtype=F32, device=dev)
            nkw = (Mk + 63) // 64
            pgkv = torch.empty(nkw * 2 * D, dtype=F32, device=dev)
            a3 = L.SaLayerBwd()
            a3.M, a3.D, a3.hidden = Mk, D, Hd
            a3.dqkv, a3.WqkvT, a3.base, a3.mean1, a3.rstd1, a3.ln1_g = (dkv.data_ptr(), packed[0]["WkvT"].data_ptr(), xkv.data_ptr(), mk.data_ptr(),
                                                                     rk.data_ptr(), lnkv.weight.data.data_ptr())
            a3.dx1, a3.dbase, a3.pgrad1, a3.dsum = None, dxkv.data_ptr(), pgkv.data_ptr(), None
            L.call_struct("vpf_ca_kv_bwd", a3)
            pgrad_job(0, 0, lnkv, partials=pgkv, rows=nkw)
            dxkv = dxkv.view(B, Lkv, D) if ctx.needs_input_grad[2] else None
        else:
            dnk = linear_dgrad(dkv, w16[D * D:], 2 * D, D)
            if ctx.needs_input_grad[2]:
                dxkv = layernorm_bwd(dnk, xkv, mk, rk, lnkv.weight, lnkv.bias, None, out_h16=(ctx.xkv_dtype == H16)).view(B, Lkv, D)
            else:
                layernorm_bwd(dnk, xkv, mk, rk, lnkv.weight, lnkv.bias, None, out_h16=True)
        if front_rows:
            # dq . Wq -> q LayerNorm' -> + dx1 -> dx, the positional-gradient sum and the LayerNorm's parameter-gradient partials in one
            # kernel (vpf_ca_front_bwd) instead of a GEMM, two LayerNorm launches and an add
            dx = torch.empty(M, D, dtype=F32, device=dev)
            a2 = L.SaLayerBwd()
            a2.M, a2.D, a2.hidden = M, D, Hd
            a2.dqkv, a2.WqkvT, a2.base, a2.mean1, a2.rstd1, a2.ln1_g = (dq.data_ptr(), packed[0]["WqT"].data_ptr(), base_ca.data_ptr(), mq.data_ptr(),
                                                                     rq.data_ptr(), lnq.weight.data.data_ptr())
            a2.dx1, a2.dbase, a2.pgrad1 = dx1.data_ptr(), dx.data_ptr(), pg[0, 0].data_ptr()
            a2.dsum, a2.dsum_init = (dsum.data_ptr() if want_pos else None), int(not dsum_started)
            L.call_struct("vpf_ca_front_bwd", a2)
            pgrad_job(0, 0, lnq)
            if npj:
                L.call_struct("vpf_ln_pgrad_reduce", pjobs, npj)
        else:
            dx = layernorm_bwd(dnq, base_ca, mq, rq, lnq.weight, lnq.bias, dx1).view(M, D)
        dpos = None
        if want_pos:
            if front_rows:
                pass                                                    # (dsum already holds the sum)
            elif dsum_started:
                dsum.add_(dx)
            else:
                dsum = dx.clone()
            if ctx.pos_shape[0] == B or B == 1:
                dpos = dsum.view(ctx.pos_shape)
            else:
                dpos = torch.zeros(ctx.pos_shape, dtype=F32, device=dev)
                L.call("vpf_rowsum_mod_f32", dsum, M, D, Lq, dpos)
        if wg_all is not None:
            wg_all.flush()                                             # behind the dgrad chain: what follows this node waits for dx, not for the weight gradients
        return (dx.view(B, Lq, D), dpos, dxkv, None, None, None, None, None) + (None,) * ctx.nparams


class AdapterKVFn(torch.autograd.Function):
    """PointCloudInputAdapter (classifier.py:31-50) + the cross-attention kv LayerNorm and K / V projections
    (partseg.py:48-51,100-116) as ONE forward kernel (vpf_adapter_kv_fwd): pts [B,N,C] -> kv h16 [B,N,2D].
    Backward: the block-by-block kernels on what the forward saved."""

    @staticmethod
    def forward(ctx, pts, adapter, cross, *params):
        ctx.nparams, ctx.params = len(params), params
        B, N, C = pts.shape
        dev = pts.device
        x = pts.contiguous().float().view(-1, C)
        M = x.shape[0]
        l0, ln, l3 = adapter.point_mlp[0], adapter.point_mlp[1], adapter.point_mlp[3]
        D = l3.weight.shape[0]                                                # 256, or 384 (BASELINE config 4: the row-block kernels of sa_rows.hip)
        catt, lnkv = cross.attention, cross.kv_norm
        qkvw = [catt.q_proj.weight, catt.k_proj.weight, catt.v_proj.weight]
        pack_params(qkvw)
        w16 = shadow(qkvw)
        pk = getattr(adapter, "_vpf_packed_kv", None)
        n2, nkv = D * 64, 2 * D * D
        if pk is None or pk.device != dev:
            pk = torch.empty(2 * (n2 + nkv), dtype=H16, device=dev)        # W2 | Wkv | W2T | WkvT  (fragment order)
            adapter._vpf_packed_kv = pk
        jobs = (L.PackJob * 64)()
        w2 = shadow([l3.weight])
        for i, (src, off, rows_, depth_, tr) in enumerate(((w2, 0, D, 64, 0), (w16[D * D:], n2, 2 * D, D, 0),
                                                           (w2, n2 + nkv, 64, D, 1), (w16[D * D:], 2 * n2 + nkv, D, 2 * D, 1))):
            jobs[i].src, jobs[i].dst, jobs[i].N, jobs[i].K, jobs[i].transposed = src.data_ptr(), pk[off:].data_ptr(), rows_, depth_, tr
        L.call_struct("vpf_pack_wfrag", jobs, 4)
        a1 = torch.empty(M, 64, dtype=H16, device=dev)
        xkv = torch.empty(M, D, dtype=H16, device=dev)
        mk = torch.empty(M, dtype=F32, device=dev)
        rk = torch.empty(M, dtype=F32, device=dev)
        nk = torch.empty(M, D, dtype=H16, device=dev)
        kv = torch.empty(M, 2 * D, dtype=H16, device=dev)
        a = L.AdapterKv()
        a.M, a.C, a.D = M, C, D
        a.x, a.W1, a.b1, a.ln_g, a.ln_b = x.data_ptr(), l0.weight.data.data_ptr(), l0.bias.data.data_ptr(), ln.weight.data.data_ptr(), ln.bias.data.data_ptr()
        a.W2, a.b2, a.lnkv_g, a.lnkv_b, a.Wkv = pk.data_ptr(), l3.bias.data.data_ptr(), lnkv.weight.data.data_ptr(), lnkv.bias.data.data_ptr(), pk[n2:].data_ptr()
        a.a1, a.xkv, a.mean, a.rstd, a.nk, a.kv = a1.data_ptr(), xkv.data_ptr(), mk.data_ptr(), rk.data_ptr(), nk.data_ptr(), kv.data_ptr()
        L.call_struct("vpf_adapter_kv_fwd", a)
        ctx.mods = (adapter, cross)
        ctx.packed = pk
        ctx.save_for_backward(x, a1, xkv, mk, rk, nk)
        return kv.view(B, N, 2 * D)

    @staticmethod
    @_sinked
    def backward(ctx, dkv):
        # Nothing downstream waits for this node (its inputs are the raw points; its results are weight gradients the kernels write
        # themselves), but autograd runs it at the END of the point-cloud branch's backward -- it was created first -- where the other
        # branch's stream has long run dry.  A two-stream trainer opens a deferral (cfg.kv_bwd_defer, a KvBwdDeferral): the ~285 us chain
        # is then issued by drain() on that stream, gated behind Group2Emb's two big backward kernels (which own every CU's register
        # file: nothing runs beside them), so it runs beside the small launches that follow them instead of behind those.
        saved = ctx.saved_tensors
        args = (ctx.mods, ctx.packed, ctx.nparams, saved, dkv)
        if cfg.kv_bwd_defer is not None and dkv.is_cuda and cfg.kv_bwd_defer.take(args):
            return (None, None, None) + (None,) * ctx.nparams
        return AdapterKVFn._backward(*args)

    @staticmethod
    def _backward(mods, packed, nparams, saved, dkv):
        x, a1, xkv, mk, rk, nk = saved
        adapter, cross = mods
        l0, ln, l3 = adapter.point_mlp[0], adapter.point_mlp[1], adapter.point_mlp[3]
        catt, lnkv = cross.attention, cross.kv_norm
        D = l3.weight.shape[0]
        M, C = x.shape
        qkvw = [catt.q_proj.weight, catt.k_proj.weight, catt.v_proj.weight]
        w16, gKV = shadow(qkvw), packed_grad(qkvw[1:])
        dkv = to_h16(dkv).view(M, 2 * D)
        if cfg.adapter_kv_bwd_fused:
            pk = packed
            n2, nkv = D * 64, 2 * D * D
            fn = L.lib().vpf_adapter_kv_pgrad_rows
            fn.argtypes, fn.restype = [ctypes.c_long, ctypes.c_int], ctypes.c_int
            nwg = fn(M, D)
            dy16 = torch.empty(M, D, dtype=H16, device=x.device)
            pg = torch.empty(nwg * 2 * D, dtype=F32, device=x.device)
            da = torch.empty(M, 64, dtype=H16, device=x.device)
            a = L.AdapterKvBwd()
            a.M, a.C, a.D = M, C, D
            a.dkv, a.WkvT, a.xkv, a.mean, a.rstd, a.lnkv_g = dkv.data_ptr(), pk[2 * n2 + nkv:].data_ptr(), xkv.data_ptr(), mk.data_ptr(), rk.data_ptr(), lnkv.weight.data.data_ptr()
            a.W2T = pk[n2 + nkv:].data_ptr()
            a.dxkv, a.da1, a.pgrad_kv = dy16.data_ptr(), da.data_ptr(), pg.data_ptr()
            L.call_struct("vpf_adapter_kv_bwd", a)
            ws = torch.empty(2048 * 64 * 11, dtype=F32, device=x.device)
            L.call("vpf_adapter_front_bwd", x, da, M, C, l0.weight.data, l0.bias.data, ln.weight.data, ln.bias.data,
                   grad_buf(l0.weight), grad_buf(l0.bias), grad_buf(ln.weight), grad_buf(ln.bias), ws, ws.numel())
            pj = (L.PgradJob * 32)()
            pj[0].partials, pj[0].rows, pj[0].dgamma, pj[0].dbeta = pg.data_ptr(), nwg, grad_buf(lnkv.weight).data_ptr(), grad_buf(lnkv.bias).data_ptr()
            pj[0].D = D
            L.call_struct("vpf_ln_pgrad_reduce", pj, 1)
            wg = WgradBatch()
            wg.add(dkv, nk, 2 * D, D, gKV)
            wg.add(dy16, a1, D, 64, grad_buf(l3.weight), grad_buf(l3.bias))
            wg.flush()
            return (None, None, None) + (None,) * nparams
        linear_wgrad(dkv, nk, 2 * D, D, gKV)
        dnk = linear_dgrad(dkv, w16[D * D:], 2 * D, D)
        dy16 = layernorm_bwd(dnk, xkv, mk, rk, lnkv.weight, lnkv.bias, None, out_h16=True).view(M, D)
        linear_wgrad(dy16, a1, D, 64, grad_buf(l3.weight), grad_buf(l3.bias))
        da = linear_dgrad(dy16, shadow([l3.weight]), D, 64)
        ws = torch.empty(2048 * 64 * 11, dtype=F32, device=x.device)
        L.call("vpf_adapter_front_bwd", x, da, M, C, l0.weight.data, l0.bias.data, ln.weight.data, ln.bias.data,
               grad_buf(l0.weight), grad_buf(l0.bias), grad_buf(ln.weight), grad_buf(ln.bias), ws, ws.numel())
        return (None, None, None) + (None,) * nparams




def adapter_kv_supported(adapter, pts) -> bool:
    if not cfg.adapter_kv_fused or pts.dim() != 3 or pts.shape[-1] > 8:
        return False
    l3 = adapter.point_mlp[3]
    return tuple(l3.weight.shape) in ((256, 64), (384, 64))


def encoder_fused_supported(ca, layers, x, xkv) -> bool:
    if not (cfg.sa_fused and cfg.enc_fused) or not sa_stack_supported(layers, x) or xkv is None or xkv.dim() != 3:
        return False
    if not getattr(ca, "attention_residual", False):
        return False
    att, mlp = ca[0].module.attention, ca[1].module
    D = x.shape[-1]
    if (att.num_heads, mlp[1].weight.shape[0]) not in FUSED_WIDTHS[D] or xkv.shape[-1] not in (D, 2 * D):
        return False
    if layers and mlp[1].weight.shape[0] != layers[0][1].module[1].weight.shape[0]:
        return False                                                   # (one MLP width per stack)
    return isinstance(ca[0].drop_path, torch.nn.Identity) and isinstance(ca[1].drop_path, torch.nn.Identity)




# --------------------------------------------------------------------------- generic dropout + residual (Residual fallback)
class DropoutAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, res, p, site):
        y16 = to_h16(y)
        res = res.contiguous().float()
        out = torch.empty(res.shape, dtype=F32, device=res.device)
        ctx.rng_st = rng.acquire(res.device, p > 0.0)
        L.call("vpf_dropout_add_fwd", y16, res, out, out.numel(), ctx.rng_st, site, float(p))
        ctx.p, ctx.site, ctx.ydt = p, site, y.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous().float()
        dy = torch.empty(dout.shape, dtype=H16, device=dout.device)
        L.call("vpf_dropout_bwd", dout, dy, dout.numel(), ctx.rng_st, ctx.site, float(ctx.p))
        return (dy if ctx.ydt == H16 else to_f32(dy)), dout, None, None


# --------------------------------------------------------------------------- BatchNorm helper (channels-last [M,C])
def _bn_stat(x, C, bn, training):
    """Returns stat [2C] = mean | rstd (batch statistics + running update when training)."""
    M = x.numel() // C
    dev = x.device
    stat = torch.empty(2 * C, dtype=F32, device=dev)
    if training:
        sums = torch.zeros(2 * C, dtype=F32, device=dev)
        colsum(x, C, sums[:C], sums[C:])
        L.call("vpf_bn_finalize", sums[:C], sums[C:], M, C, float(bn.eps), float(bn.momentum), 1, bn.running_mean,
               bn.running_var, bn.num_batches_tracked, stat)
    else:
        L.call("vpf_bn_finalize", None, None, M, C, float(bn.eps), float(bn.momentum), 0, bn.running_mean, bn.running_var,
               None, stat)
    return stat


def _bn_act(x, C, stat, bn, relu, out_h16):
    M = x.numel() // C
    y = torch.empty(M, C, dtype=H16 if out_h16 else F32, device=x.device)
    L.call("vpf_bn_act_fwd", x, int(x.dtype == H16), stat, bn.weight.data, bn.bias.data, y, int(out_h16), M, C, int(relu))
    return y


def _bn_bwd(dy, x, C, stat, bn, relu, training, out_h16, want_dx=True):
    M = x.numel() // C
    tmp = torch.zeros(2 * C, dtype=F32, device=x.device)
    dx = torch.empty(M, C, dtype=H16 if out_h16 else F32, device=x.device) if want_dx else None
    L.call("vpf_bn_bwd", dy, int(dy.dtype == H16), x, int(x.dtype == H16), stat, bn.weight.data, bn.bias.data, M, C,
           int(relu), int(training), tmp, dx, int(out_h16), grad_buf(bn.weight), grad_buf(bn.bias))
    return dx


# --------------------------------------------------------------------------- Group2Emb (utils.py:144-189)
G2E_DEBUG = {}      # {"dbg": int64 tensor [256*2*6]} -> per-phase cycle stamps of vpf_g2e_bwd (diagnostic)




class Group2EmbFn(torch.autograd.Function):
    """conv(C,64) BN ReLU conv(64,128) -> max over K -> cat[global, local] -> conv(256,256) BN ReLU conv(256,D) -> max over K.

    The concatenation is never formed: conv(256,256) on [global | local] = local . W[:,128:]^T + a per-group
    bias (global . W[:,:128]^T + b), which also removes a quarter of the block's MACs.  BatchNorm-1 statistics
    come from the 3x3 second-moment matrix of the inputs (the first conv is affine in x)."""

    @staticmethod
    def forward(ctx, groups, mod, training, *params):
        ctx.nparams, ctx.params = len(params), params
        Bq, G, K, C = groups.shape
        dev = groups.device
        x = groups.contiguous().float().view(-1, C)
        M, NG = x.shape[0], Bq * G
        c1, bn1, c2 = mod.first_conv[0], mod.first_conv[1], mod.first_conv[3]
        c3, bn2, c4 = mod.second_conv[0], mod.second_conv[1], mod.second_conv[3]
        Dm = c4.weight.shape[0]
        w1 = c1.weight.data.view(64, C)
        stat1 = torch.empty(128, dtype=F32, device=dev)
        scratch = None
        fast = K == 32 and C == 3 and Dm % 32 == 0 and Dm <= 512
        merged = training and fast and cfg.g2e_bn_merged
        if merged:
            # moments -> statistics -> affine -> folded conv in two launches (vpf_g2e_bn1_prepare)
            scratch = torch.empty(72 + 512 * 72, dtype=F32, device=dev)
            ab1 = torch.empty(128, dtype=F32, device=dev)
            w1e = torch.empty(64 * C + 64, dtype=F32, device=dev)
            L.call("vpf_g2e_bn1_prepare", x, M, C, w1, c1.bias.data, scratch, bn1.weight.data, bn1.bias.data, float(bn1.eps),
                   float(bn1.momentum), bn1.running_mean, bn1.running_var, bn1.num_batches_tracked, stat1, ab1, w1e, w1e[64 * C:])
        elif training:
            scratch = torch.empty(72 + 512 * 72, dtype=F32, device=dev)
            sums = torch.empty(128, dtype=F32, device=dev)
            L.call("vpf_g2e_conv1_stats_moments", x, M, C, w1, c1.bias.data, scratch, sums[:64], sums[64:])
            L.call("vpf_bn_finalize", sums[:64], sums[64:], M, 64, float(bn1.eps), float(bn1.momentum), 1, bn1.running_mean,
                   bn1.running_var, bn1.num_batches_tracked, stat1)
        else:
            L.call("vpf_bn_finalize", None, None, M, 64, float(bn1.eps), float(bn1.momentum), 0, bn1.running_mean,
                   bn1.running_var, None, stat1)
        if fast:
            # two persistent weight-stationary kernels: activations of a pair of groups never leave the CU except h3
            if not merged:
                ab1 = torch.empty(128, dtype=F32, device=dev)
                L.call("vpf_bn_affine", stat1, bn1.weight.data, bn1.bias.data, 64, ab1)
                w1e = torch.empty(64 * C + 64, dtype=F32, device=dev)
                L.call("vpf_g2e_fold_bn1", w1, c1.bias.data, ab1, C, w1e, w1e[64 * C:])
            a1 = torch.empty(M, 64, dtype=H16, device=dev)
            h2 = torch.empty(M, 128, dtype=H16, device=dev)
            gmax = torch.empty(NG, 128, dtype=H16, device=dev)
            arg2 = torch.empty(NG, 128, dtype=torch.uint8, device=dev)
            h3 = torch.empty(M, 256, dtype=H16, device=dev)
            part = torch.empty(256 * 512, dtype=F32, device=dev)
            nwg = ctypes.c_int(0)
            L.call("vpf_g2e_fwd_a", x, NG, C, w1e, w1e[64 * C:], shadow([c2.weight]), c2.bias.data, shadow([c3.weight]), c3.bias.data,
                   a1, h2, gmax, arg2, h3, part, ctypes.addressof(nwg))
            stat2 = torch.empty(512, dtype=F32, device=dev)
            ab2 = torch.empty(512, dtype=F32, device=dev)
            if merged:
                # fixed-order fold of the per-workgroup partials (deterministic BatchNorm-2 statistics) + statistics + affine: one launch
                L.call("vpf_bn_partials_finalize", part, nwg.value, 256, M, bn2.weight.data, bn2.bias.data, float(bn2.eps),
                       float(bn2.momentum), bn2.running_mean, bn2.running_var, bn2.num_batches_tracked, stat2, ab2)
            else:
                if training:
                    sums2 = torch.empty(512, dtype=F32, device=dev)
                    L.call("vpf_sum_rows_f32", part, nwg.value, 512, sums2)
                    L.call("vpf_bn_finalize", sums2[:256], sums2[256:], M, 256, float(bn2.eps), float(bn2.momentum), 1, bn2.running_mean,
                           bn2.running_var, bn2.num_batches_tracked, stat2)
                else:
                    L.call("vpf_bn_finalize", None, None, M, 256, float(bn2.eps), float(bn2.momentum), 0, bn2.running_mean,
                           bn2.running_var, None, stat2)
                L.call("vpf_bn_affine", stat2, bn2.weight.data, bn2.bias.data, 256, ab2)
            out = torch.empty(NG, Dm, dtype=F32, device=dev)
            arg4 = torch.empty(NG, Dm, dtype=torch.uint8, device=dev)
            L.call("vpf_g2e_fwd_b", h3, NG, ab2, shadow([c4.weight]), c4.bias.data, Dm, out, arg4)
            fused = True
        else:
            a1 = torch.empty(M, 64, dtype=H16, device=dev)
            L.call("vpf_g2e_conv1_apply", x, M, C, w1, c1.bias.data, stat1, bn1.weight.data, bn1.bias.data, a1)
            h2 = linear_fwd(a1, shadow([c2.weight]), 128, 64, c2.bias.data)                      # [M,128]
            gmax = torch.empty(NG, 128, dtype=H16, device=dev)
            arg2 = torch.empty(NG, 128, dtype=torch.uint8, device=dev)
            L.call("vpf_group_max_fwd", h2, NG, K, 128, gmax, 1, arg2)
            w3 = shadow([c3.weight])                                                              # [256, 256] = [global | local]
            gb = torch.empty(NG, 256, dtype=F32, device=dev)
            gemm(gmax, 0, 128, w3, 0, 256, NG, 256, 128, gb, 256, c_f32=True, bias=c3.bias.data)  # global . W[:, :128]^T + b
            h3 = torch.empty(M, 256, dtype=H16, device=dev)
            gemm(h2, 0, 128, w3[128:], 0, 256, M, 256, 128, h3, 256, c_f32=False, mode=EPI_GROUPBIAS, gbias=gb, group=K)
            stat2 = _bn_stat(h3, 256, bn2, training)
            ab2 = torch.empty(512, dtype=F32, device=dev)
            L.call("vpf_bn_affine", stat2, bn2.weight.data, bn2.bias.data, 256, ab2)
            out = torch.empty(NG, Dm, dtype=F32, device=dev)
            arg4 = torch.empty(NG, Dm, dtype=torch.uint8, device=dev)
            fused = (32 % K == 0) and (Dm % 8 == 0)
            if fused:
                gemm_fused(h3, 0, 256, shadow([c4.weight]), 0, 256, M, Dm, 256, out, Dm, c_f32=True, mode=EPI_GROUPMAX, a_kind=1,
                           a_ab=ab2, bias=c4.bias.data, C2=arg4, ldc2=Dm, group=K)
            else:
                a3 = _bn_act(h3, 256, stat2, bn2, True, True)
                h4 = linear_fwd(a3, shadow([c4.weight]), Dm, 256, c4.bias.data)
                L.call("vpf_group_max_fwd", h4, NG, K, Dm, out, 0, arg4)
        ctx.mod, ctx.training, ctx.dims, ctx.fused = mod, training, (Bq, G, K, C, Dm), fused
        ctx.mom = scratch[:72] if scratch is not None else None          # sum x | sum x x^T: the one-pass BatchNorm-1 backward needs them
        ctx.save_for_backward(x, stat1, a1, arg2, h2, gmax, h3, stat2, ab2, arg4)
        return out.view(Bq, G, Dm)

    @staticmethod
    @_sinked
    def backward(ctx, dout):
        x, stat1, a1, arg2, h2, gmax, h3, stat2, ab2, arg4 = ctx.saved_tensors
        mod, training = ctx.mod, ctx.training
        Bq, G, K, C, Dm = ctx.dims
        dev = dout.device
        M, NG = x.shape[0], Bq * G
        c1, bn1, c2 = mod.first_conv[0], mod.first_conv[1], mod.first_conv[3]
        c3, bn2, c4 = mod.second_conv[0], mod.second_conv[1], mod.second_conv[3]
        dout = dout.contiguous().float()
        w3 = shadow([c3.weight])
        gW3 = grad_buf(c3.weight).view(256, 256)
        if ctx.fused and K == 32 and C == 3 and Dm <= 256 and Dm % 16 == 0:
            # persistent fused path: sparse wgrad of the last conv, then dgrad + BatchNorm-2 backward + conv3 dgrad on MFMA
            # with transposed weight fragments in registers; only dh3 / dh2 reach HBM
            w4t = torch.empty(256 * Dm, dtype=H16, device=dev)
            L.call("vpf_transpose_h16", shadow([c4.weight]), 256, Dm, 256, w4t)
            w3bt = torch.empty(128 * 256, dtype=H16, device=dev)
            L.call("vpf_transpose_h16", w3[128:], 256, 256, 128, w3bt)
            L.call("vpf_g2e_wgrad4", h3, NG, ab2, dout, arg4, Dm, grad_buf(c4.weight), grad_buf(c4.bias))
            tmp2 = torch.zeros(512, dtype=F32, device=dev)
            dh3 = torch.empty(M, 256, dtype=H16, device=dev)
            dgb = torch.empty(NG, 256, dtype=F32, device=dev)
            dh2 = torch.empty(M, 128, dtype=H16, device=dev)
            L.call("vpf_g2e_bwd", dout, arg4, Dm, NG, h3, stat2, bn2.weight.data, bn2.bias.data, w4t, w3bt, int(training), tmp2, dh3,
                   dgb, dh2, grad_buf(bn2.weight), grad_buf(bn2.bias), G2E_DEBUG.get("dbg"))
            if cfg.kv_bwd_defer is not None:
                cfg.kv_bwd_defer.mark_gate()
        else:
            if ctx.fused:
                # d(conv output) = max-pool gradient: rebuilt on the fly from (dout, arg4) inside both GEMMs' A-operand
                # loads; relu(bn(h3)) rebuilt inside the wgrad's B-operand load
                if K == 32:
                    # the sparse walk (one non-zero per (group, column)) has no width limit: 32 x less work than the dense product
                    # (config 4, Dm = 384: 234 -> ~70 us)
                    L.call("vpf_g2e_wgrad4", h3, NG, ab2, dout, arg4, Dm, grad_buf(c4.weight), grad_buf(c4.bias))
                else:
                    gemm_fused(dout, 1, Dm, h3, 1, 256, Dm, 256, M, grad_buf(c4.weight), 256, c_f32=True, mode=EPI_ATOMIC, a_kind=2,
                               a_dout=dout, a_arg=arg4, a_group=K, a_ncols=Dm, b_kind=1, b_ab=ab2, dbias=grad_buf(c4.bias))
                da3 = torch.empty(M, 256, dtype=H16, device=dev)
                gemm_fused(dout, 0, Dm, shadow([c4.weight]), 1, 256, M, 256, Dm, da3, 256, c_f32=False, a_kind=2, a_dout=dout,
                           a_arg=arg4, a_group=K, a_ncols=Dm)
            else:
                a3 = _bn_act(h3, 256, stat2, bn2, True, True)
                dh4 = torch.empty(M, Dm, dtype=H16, device=dev)
                L.call("vpf_group_max_bwd", dout, 0, arg4, NG, K, Dm, dh4)
                linear_wgrad(dh4, a3, Dm, 256, grad_buf(c4.weight), grad_buf(c4.bias))
                da3 = linear_dgrad(dh4, shadow([c4.weight]), Dm, 256)
            dh3 = _bn_bwd(da3, h3, 256, stat2, bn2, True, training, True)
            # conv(256,256) on [global | local]: per-group part and per-point part
            dgb = torch.empty(NG, 256, dtype=F32, device=dev)
            L.call("vpf_group_sum", dh3, NG, K, 256, dgb)                                        # d(per-group bias)
            dh2 = torch.empty(M, 128, dtype=H16, device=dev)
            gemm(dh3, 0, 256, w3[128:], 1, 256, M, 128, 256, dh2, 128, c_f32=False)                            # dlocal
        # (round 5, measured and taken back: colsum / cast / BatchNorm-2's parameter gradients from inside vpf_g2e_bwd -- three launches
        #  fewer, +0.01 ms per step: NOTES.md)
        colsum(dgb, 256, grad_buf(c3.bias))
        dgb16 = to_h16(dgb)
        gemm(dgb16, 1, 256, gmax, 1, 128, 256, 128, NG, gW3, 256, c_f32=True, mode=EPI_ATOMIC)            # dW[:, :128]
        gemm(dh3, 1, 256, h2, 1, 128, 256, 128, M, gW3[:, 128:], 256, c_f32=True, mode=EPI_ATOMIC)        # dW[:, 128:]
        dgmax = torch.empty(NG, 128, dtype=H16, device=dev)
        gemm(dgb16, 0, 256, w3, 1, 256, NG, 128, 256, dgmax, 128, c_f32=False)                             # dglobal
        L.call("vpf_group_max_scatter_add", dgmax, arg2, NG, K, 128, dh2)
        linear_wgrad(dh2, a1, 128, 64, grad_buf(c2.weight), grad_buf(c2.bias))
        ws = torch.empty(1025 * 320, dtype=F32, device=dev)                      # per-block partial sums (no atomics)
        if cfg.g2e_conv1_bwd_fused:
            # conv2's input gradient is formed and consumed inside the first conv's backward: da1 [M, 64] never exists
            L.call("vpf_g2e_conv1_bwd_fused", x, dh2, M, C, c1.weight.data.view(64, C), c1.bias.data, stat1, bn1.weight.data, bn1.bias.data,
                   int(training), ctx.mom, shadow([c2.weight]), grad_buf(c1.weight), grad_buf(c1.bias), grad_buf(bn1.weight),
                   grad_buf(bn1.bias), ws, ws.numel())
        else:
            da1 = linear_dgrad(dh2, shadow([c2.weight]), 128, 64)
            L.call("vpf_g2e_conv1_bwd", x, da1, M, C, c1.weight.data.view(64, C), c1.bias.data, stat1, bn1.weight.data, bn1.bias.data,
                   int(training), ctx.mom, grad_buf(c1.weight), grad_buf(c1.bias), grad_buf(bn1.weight), grad_buf(bn1.bias), ws, ws.numel())
        return (None, None, None) + (None,) * ctx.nparams


# --------------------------------------------------------------------------- adapter / position MLP / patch embedding
class AdapterFn(torch.autograd.Function):
    """PointCloudInputAdapter.point_mlp (classifier.py:31-36): Linear(C,64) LN ReLU Linear(64,D) -> h16 [B,N,D]."""

    @staticmethod
    def forward(ctx, pts, mod, *params):
        ctx.nparams, ctx.params = len(params), params
        B, N, C = pts.shape
        x = pts.contiguous().float().view(-1, C)
        M = x.shape[0]
        l0, ln, l3 = mod.point_mlp[0], mod.point_mlp[1], mod.point_mlp[3]
        D = l3.weight.shape[0]
        a = torch.empty(M, 64, dtype=H16, device=x.device)
        L.call("vpf_adapter_front_fwd", x, M, C, l0.weight.data, l0.bias.data, ln.weight.data, ln.bias.data, a)
        y = linear_fwd(a, shadow([l3.weight]), D, 64, l3.bias.data)
        ctx.mod = mod
        ctx.save_for_backward(x, a)
        return y.view(B, N, D)

    @staticmethod
    @_sinked
    def backward(ctx, dy):
        x, a = ctx.saved_tensors
        mod = ctx.mod
        l0, ln, l3 = mod.point_mlp[0], mod.point_mlp[1], mod.point_mlp[3]
        D = l3.weight.shape[0]
        M, C = x.shape
        dy16 = to_h16(dy).view(M, D)
        linear_wgrad(dy16, a, D, 64, grad_buf(l3.weight), grad_buf(l3.bias))
        da = linear_dgrad(dy16, shadow([l3.weight]), D, 64)
        ws = torch.empty(2048 * 64 * 11, dtype=F32, device=x.device)          # per-block partial parameter gradients
        L.call("vpf_adapter_front_bwd", x, da, M, C, l0.weight.data, l0.bias.data, ln.weight.data, ln.bias.data,
               grad_buf(l0.weight), grad_buf(l0.bias), grad_buf(ln.weight), grad_buf(ln.bias), ws, ws.numel())
        return (None, None) + (None,) * ctx.nparams


class PosMLPFn(torch.autograd.Function):
    """position_emb (partseg.py:498-501): Linear(3,128) GELU Linear(128,D) -> f32 [B,G,D]."""

    @staticmethod
    def forward(ctx, centers, seq, *params):
        ctx.nparams, ctx.params = len(params), params
        B, G, C = centers.shape
        x = centers.contiguous().float().view(-1, C)
        M = x.shape[0]
        l0, l2 = seq[0], seq[2]
        Hd, D = l0.weight.shape[0], l2.weight.shape[0]
        g = torch.empty(M, Hd, dtype=H16, device=x.device)
        L.call("vpf_smallk_fwd", x, M, C, l0.weight.data, l0.bias.data, Hd, 1, g)
        y = linear_fwd(g, shadow([l2.weight]), D, Hd, l2.bias.data, out_f32=True)
        ctx.seq = seq
        ctx.save_for_backward(x, g)
        return y.view(B, G, D)

    @staticmethod
    @_sinked
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        l0, l2 = ctx.seq[0], ctx.seq[2]
        Hd, D = l0.weight.shape[0], l2.weight.shape[0]
        M, C = x.shape
        dy16 = to_h16(dy).view(M, D)
        linear_wgrad(dy16, g, D, Hd, grad_buf(l2.weight), grad_buf(l2.bias))
        dg = linear_dgrad(dy16, shadow([l2.weight]), D, Hd)
        L.call("vpf_smallk_bwd", x, dg, M, C, l0.weight.data, l0.bias.data, Hd, 1, grad_buf(l0.weight), grad_buf(l0.bias))
        return (None, None) + (None,) * ctx.nparams


class CaFrontFn(torch.autograd.Function):
    """position_emb AND the front of the cross-attention layer in one kernel (vpf_ca_front_fwd): pos = position_emb(centres) as
    PosMLPFn computes it, plus -- for the encoder call that follows -- base = tokens + pos, q_norm(base) and the q projection, left
    on the cross-attention layer as a one-shot stash that EncoderFusedFn.forward picks up instead of launching LayerNorm + GEMM
    itself (together with the fragment-order weight copies of the whole encoder, which this call makes: one pack launch, as before).
    Autograd sees exactly PosMLPFn: the tokens enter detached (their gradient flows through the encoder Function, which saves the
    stashed tensors like its own), backward is PosMLPFn.backward."""

    @staticmethod
    def forward(ctx, centers, seq, tokens, enc, *params):
        ctx._vpf_iscale = _current_iscale()                 # (backward is PosMLPFn.backward: the sink reads the scale from THIS ctx)
        ctx.nparams, ctx.params = len(params), params
        B, G, C = centers.shape
        x = centers.contiguous().float().view(-1, C)
        M = x.shape[0]
        l0, l2 = seq[0], seq[2]
        Hd, D = l0.weight.shape[0], l2.weight.shape[0]
        dev = x.device
        ca, layers = enc.cross_attn_1, enc.sa_layers
        cross = ca[0].module
        catt, lnq = cross.attention, cross.q_norm
        blocks = [(catt, ca[1].module, False, False)] + [(l[0].module.attention, l[1].module, True, True) for l in layers]
        packed = _pack_blocks(blocks, ca, dev, front=l2)
        tok = tokens.contiguous().float()
        hpos = torch.empty(M, Hd, dtype=H16, device=dev)
        pos = torch.empty(M, D, dtype=F32, device=dev)
        base = torch.empty(M, D, dtype=F32, device=dev)
        mq, rq = torch.empty(M, dtype=F32, device=dev), torch.empty(M, dtype=F32, device=dev)
        nq, q = torch.empty(M, D, dtype=H16, device=dev), torch.empty(M, D, dtype=H16, device=dev)
        a = L.CaFront()
        a.M, a.D, a.hidden, a.C = M, D, Hd, C
        a.centers, a.W0, a.b0 = x.data_ptr(), l0.weight.data.data_ptr(), l0.bias.data.data_ptr()
        a.W1, a.b1 = packed[0]["Wpos"].data_ptr(), l2.bias.data.data_ptr()
        a.x, a.lnq_g, a.lnq_b, a.Wq = tok.data_ptr(), lnq.weight.data.data_ptr(), lnq.bias.data.data_ptr(), packed[0]["Wq"].data_ptr()
        a.hpos, a.pos, a.base, a.mean, a.rstd, a.nq, a.q = (hpos.data_ptr(), pos.data_ptr(), base.data_ptr(), mq.data_ptr(), rq.data_ptr(),
                                                              nq.data_ptr(), q.data_ptr())
        L.call_struct("vpf_ca_front_fwd", a)
        # (the stash HOLDS tok / pos: a recycled allocation can then never alias the pointers it is matched by)
        ca._vpf_front_stash = dict(tokens=tok.data_ptr(), pos=pos.data_ptr(), shape=(B, G, D), base=base, mq=mq, rq=rq, nq=nq, q=q,
                                   packed=packed, nblocks=len(blocks), keep=(tok, pos))
        ctx.seq = seq
        ctx.save_for_backward(x, hpos)
        return pos.view(B, G, D)

    @staticmethod
    def backward(ctx, dy):
        g = PosMLPFn.backward(ctx, dy)                     # (None, None, *parameter gradients): the same kernels, the same saved tensors
        return g[:2] + (None, None) + g[2:]                # + the detached tokens and the encoder object


def ca_front_supported(seq, tokens, enc) -> bool:
    """vpf_ca_front_fwd: D = 256, Linear(3,128) position MLP, one cross-attention layer on the fused encoder path."""
    if not (cfg.sa_fused and cfg.enc_fused and cfg.ca_front_fused) or not tokens.is_cuda or tokens.dim() != 3:
        return False
    l0, l2 = seq[0], seq[2]
    if tokens.shape[-1] != 256 or tuple(l0.weight.shape) != (128, 3) or l2.weight.shape[1] != 128 or l2.weight.shape[0] != 256:
        return False
    if enc.num_cross_attention_layers != 1:
        return False
    # the stash is only ever consumed by EncoderFusedFn: the encoder call that follows must take the fused path
    probe = tokens.new_empty((tokens.shape[0], 1, 2 * tokens.shape[2]))
    fused_ok = getattr(enc, "fused_ok", None)              # (kernel tests hand in a bare stand-in for the encoder)
    return bool(fused_ok(tokens, probe)) if fused_ok is not None else True


class PatchEmbedFn(torch.autograd.Function):
    """patch2emb (partseg.py:631-634): 'b (h p1)(w p2) c -> b (h w)(p1 p2 c)' + Linear(3p^2, D) -> f32 [B,T,D]."""

    @staticmethod
    def forward(ctx, imgs, lin, p, *params):
        ctx.nparams, ctx.params = len(params), params
        B, Hh, Ww, C = imgs.shape
        if imgs.dtype != F32:
            imgs = imgs.float()
        T, pd = (Hh // p) * (Ww // p), p * p * C
        if Hh % p or Ww % p:
            raise L.VpfError(f"image {Hh}x{Ww} is not divisible by patch_size {p}")
        patches = torch.empty(B * T, pd, dtype=H16, device=imgs.device)
        sb, sh, sw, sc = imgs.stride()
        L.call("vpf_patchify", imgs, sb, sh, sw, sc, B, Hh, Ww, C, p, patches)
        D = lin.weight.shape[0]
        y = linear_fwd(patches, shadow([lin.weight]), D, pd, lin.bias.data, out_f32=True)
        ctx.lin = lin
        ctx.save_for_backward(patches)
        return y.view(B, T, D)

    @staticmethod
    @_sinked
    def backward(ctx, dy):
        (patches,) = ctx.saved_tensors
        lin = ctx.lin
        D, pd = lin.weight.shape
        dy16 = to_h16(dy).view(-1, D)
        linear_wgrad(dy16, patches, D, pd, grad_buf(lin.weight), grad_buf(lin.bias))
        return (None, None, None) + (None,) * ctx.nparams


# --------------------------------------------------------------------------- pooling + projection head
class PoolFn(torch.autograd.Function):
    """cat[max over tokens, mean over tokens]  (partseg.py:547)."""

    @staticmethod
    def forward(ctx, x):
        B, Lt, D = x.shape
        x = x.contiguous().float()
        out = torch.empty(B, 2 * D, dtype=F32, device=x.device)
        arg = torch.empty(B, D, dtype=torch.int32, device=x.device)
        L.call("vpf_pool_fwd", x, B, Lt, D, out, arg)
        ctx.dims = (B, Lt, D)
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, dout):
        (arg,) = ctx.saved_tensors
        B, Lt, D = ctx.dims
        dx = torch.empty(B, Lt, D, dtype=F32, device=dout.device)
        L.call("vpf_pool_bwd", dout.contiguous().float(), arg, B, Lt, D, dx)
        return dx


class HeadFn(torch.autograd.Function):
    """latent_head (partseg.py:519-525): BN1d ReLU Linear(no bias) BN1d ReLU Linear(no bias), fp32 in/out."""

    @staticmethod
    def forward(ctx, x, seq, training, *params):
        ctx.nparams, ctx.params = len(params), params
        bn1, l1, bn2, l2 = seq[0], seq[2], seq[3], seq[5]
        Bn, C1 = x.shape
        C2, C3 = l1.weight.shape[0], l2.weight.shape[0]
        x = x.contiguous().float()
        small = training and Bn <= 4096 and C1 % 64 == 0 and C2 % 64 == 0
        ctx.small = small

        def bn_relu(t, C, bn):
            if not small:
                st = _bn_stat(t, C, bn, training)
                return st, _bn_act(t, C, st, bn, True, True)
            st = torch.empty(2 * C, dtype=F32, device=t.device)
            y_ = torch.empty(Bn, C, dtype=H16, device=t.device)
            L.call("vpf_bn_small_fwd", t, Bn, C, bn.weight.data, bn.bias.data, float(bn.eps), float(bn.momentum), bn.running_mean,
                   bn.running_var, bn.num_batches_tracked, st, y_, 1)
            return st, y_

        s1, a1 = bn_relu(x, C1, bn1)
        h = linear_fwd(a1, shadow([l1.weight]), C2, C1, l1.bias.data if l1.bias is not None else None, out_f32=True)
        s2, a2 = bn_relu(h, C2, bn2)
        y = linear_fwd(a2, shadow([l2.weight]), C3, C2, l2.bias.data if l2.bias is not None else None, out_f32=True)
        ctx.seq, ctx.training = seq, training
        ctx.save_for_backward(x, s1, a1, h, s2, a2)
        return y

    @staticmethod
    @_sinked
    def backward(ctx, dy):
        x, s1, a1, h, s2, a2 = ctx.saved_tensors
        seq, training = ctx.seq, ctx.training
        bn1, l1, bn2, l2 = seq[0], seq[2], seq[3], seq[5]
        C1 = x.shape[1]
        C2, C3 = l1.weight.shape[0], l2.weight.shape[0]
        dy16 = to_h16(dy)
        if l2.bias is not None:
            colsum(dy16, C3, grad_buf(l2.bias))
        linear_wgrad(dy16, a2, C3, C2, grad_buf(l2.weight))
        da2 = linear_dgrad(dy16, shadow([l2.weight]), C3, C2, out_f32=True)

        def bn_bwd(dt, t, C, st, bn, out_h16):
            if not ctx.small:
                return _bn_bwd(dt, t, C, st, bn, True, training, out_h16)
            dx_ = torch.empty(t.shape[0], C, dtype=H16 if out_h16 else F32, device=t.device)
            L.call("vpf_bn_small_bwd", dt, t, st, bn.weight.data, bn.bias.data, t.shape[0], C, 1, dx_, int(out_h16),
                   grad_buf(bn.weight), grad_buf(bn.bias))
            return dx_

        dh = bn_bwd(da2, h, C2, s2, bn2, True)
        if l1.bias is not None:
            colsum(dh, C2, grad_buf(l1.bias))
        linear_wgrad(dh, a1, C2, C1, grad_buf(l1.weight))
        da1 = linear_dgrad(dh, shadow([l1.weight]), C2, C1, out_f32=True)
        dx = bn_bwd(da1, x, C1, s1, bn1, False)
        return (dx, None, None) + (None,) * ctx.nparams


class BnReluLinearFn(torch.autograd.Function):
    """One BatchNorm1d - ReLU - Linear block of an MLP head (finetune_head, partseg.py:572-581), fp32 in / out: the same kernels
    as HeadFn (one-kernel BatchNorm for small batches, MFMA GEMM with a fused bias) for any number of blocks."""

    @staticmethod
    def forward(ctx, x, bn, lin, training, *params):
        ctx.nparams, ctx.params = len(params), params
        Bn, C1 = x.shape
        C2 = lin.weight.shape[0]
        x = x.contiguous().float()
        small = training and Bn <= 4096 and C1 % 64 == 0
        if small:
            st = torch.empty(2 * C1, dtype=F32, device=x.device)
            a = torch.empty(Bn, C1, dtype=H16, device=x.device)
            L.call("vpf_bn_small_fwd", x, Bn, C1, bn.weight.data, bn.bias.data, float(bn.eps), float(bn.momentum), bn.running_mean,
                   bn.running_var, bn.num_batches_tracked, st, a, 1)
        else:
            st = _bn_stat(x, C1, bn, training)
            a = _bn_act(x, C1, st, bn, True, True)
        y = linear_fwd(a, shadow([lin.weight]), C2, C1, lin.bias.data if lin.bias is not None else None, out_f32=True)
        ctx.mods, ctx.training, ctx.small = (bn, lin), training, small
        ctx.save_for_backward(x, st, a)
        return y

    @staticmethod
    @_sinked
    def backward(ctx, dy):
        x, st, a = ctx.saved_tensors
        bn, lin = ctx.mods
        C1, C2 = x.shape[1], lin.weight.shape[0]
        dy16 = to_h16(dy)
        if lin.bias is not None:
            colsum(dy16, C2, grad_buf(lin.bias))
        linear_wgrad(dy16, a, C2, C1, grad_buf(lin.weight))
        da = linear_dgrad(dy16, shadow([lin.weight]), C2, C1, out_f32=True)
        if ctx.small:
            dx = torch.empty(x.shape[0], C1, dtype=F32, device=x.device)
            L.call("vpf_bn_small_bwd", da, x, st, bn.weight.data, bn.bias.data, x.shape[0], C1, 1, dx, 0, grad_buf(bn.weight), grad_buf(bn.bias))
        else:
            dx = _bn_bwd(da, x, C1, st, bn, True, ctx.training, False)
        return (dx, None, None, None) + (None,) * ctx.nparams


# --------------------------------------------------------------------------- NT-Xent
class NTXentFn(torch.autograd.Function):
    """lightly==1.1.21 NTXentLoss(temperature, memory_bank_size=0) -- pretrain.py:155,196,202."""

    @staticmethod
    def forward(ctx, z0, z1, temperature):
        b, D = z0.shape
        z0, z1 = z0.contiguous().float(), z1.contiguous().float()
        dev = z0.device
        n = 2 * b
        zn = torch.empty(n, D, dtype=F32, device=dev)
        inv = torch.empty(n, dtype=F32, device=dev)
        P = torch.empty(n, n, dtype=F32, device=dev)
        rows = torch.empty(n, dtype=F32, device=dev)
        loss = torch.empty((), dtype=F32, device=dev)
        L.call("vpf_ntxent_fwd", z0, z1, b, D, float(temperature), zn, inv, P, rows, loss)
        ctx.t, ctx.dims = temperature, (b, D)
        ctx.save_for_backward(zn, inv, P)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        zn, inv, P = ctx.saved_tensors
        b, D = ctx.dims
        dz0 = torch.empty(b, D, dtype=F32, device=zn.device)
        dz1 = torch.empty(b, D, dtype=F32, device=zn.device)
        L.call("vpf_ntxent_bwd", zn, inv, P, b, D, float(ctx.t), dloss.contiguous().float().reshape(1), dz0, dz1)
        return dz0, dz1, None


class Timeline:
    """Device timestamps at named points of the step (vpf_stamp: one lane writes the constant-rate clock).  The marks are ordinary
    launches on torch's current stream, so they are captured with the step and every replay refreshes them: `read()` after a replay
    tells when the replayed graph reached each point -- what a kernel trace cannot (it changes the overlap it observes)."""

    def __init__(self, device, slots: int = 128):
        self.buf = torch.zeros(slots, dtype=torch.int64, device=device)
        self.names = []
        fn = L.lib().vpf_wall_clock_khz
        fn.argtypes, fn.restype = [], L.I
        self.khz = int(fn())
        if self.khz <= 0:
            raise L.VpfError("vpf_wall_clock_khz failed")

    def mark(self, name: str) -> None:
        if name not in self.names:
            if len(self.names) == self.buf.numel():
                raise L.VpfError("Timeline: out of slots")
            self.names.append(name)
        L.call("vpf_stamp", self.buf, self.names.index(name))

    def read(self) -> dict:
        """{name: microseconds since the earliest mark} (call after a synchronize)."""
        v = self.buf[:len(self.names)].cpu().tolist()
        t0 = min(v)
        return {n: (t - t0) * 1e3 / self.khz for n, t in zip(self.names, v)}


class StampFn(torch.autograd.Function):
    """Identity that marks `fwd_name` in the forward pass and `bwd_name` when the gradient comes back through it."""

    @staticmethod
    def forward(ctx, x, timeline, fwd_name, bwd_name):
        ctx.timeline, ctx.bwd_name = timeline, bwd_name
        if fwd_name:
            timeline.mark(fwd_name)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if ctx.bwd_name:
            ctx.timeline.mark(ctx.bwd_name)
        return g, None, None, None


class PretrainLossFn(torch.autograd.Function):
    """pretrain.py:196-204 in three launches: returns (total = imid + w * cmid [scalar], parts f32[2] = (imid, cmid));
    only total carries a gradient (the parts are the logged values)."""

    @staticmethod
    def forward(ctx, f, g, temperature, w):
        f = f.contiguous().float(); g = g.contiguous().float()
        b, D = g.shape
        n = 2 * b
        dev = f.device
        zn = torch.empty(2, n, D, dtype=F32, device=dev)
        inv = torch.empty(2, n, dtype=F32, device=dev)
        P = torch.empty(2, n, n, dtype=F32, device=dev)
        rows = torch.empty(2, n, dtype=F32, device=dev)
        total = torch.empty((), dtype=F32, device=dev)
        parts = torch.empty(2, dtype=F32, device=dev)
        L.call("vpf_pretrain_loss_fwd", f, g, b, D, float(temperature), float(w), zn, inv, P, rows, total, parts)
        ctx.t, ctx.w, ctx.dims = temperature, w, (b, D)
        ctx.save_for_backward(zn, inv, P)
        ctx.mark_non_differentiable(parts)
        ctx.set_materialize_grads(False)            # no zero-filled gradient for `parts` (a launch per step)
        return total, parts

    @staticmethod
    def backward(ctx, dlosses, _dparts):
        zn, inv, P = ctx.saved_tensors
        b, D = ctx.dims
        dev = zn.device
        ws = torch.empty(2, 2 * b, D, dtype=F32, device=dev)
        df = torch.empty(2 * b, D, dtype=F32, device=dev)
        dg = torch.empty(b, D, dtype=F32, device=dev)
        L.call("vpf_pretrain_loss_bwd", zn, inv, P, b, D, float(ctx.t), float(ctx.w), dlosses.contiguous().float().reshape(1), ws, df, dg)
        return df, dg, None, None


def pretrain_losses(f, g, temperature=0.1, cmid_weight=1.0):
    """f [2b, D]: features of cat(view1, view2); g [b, D]: image features -> (total [scalar], parts f32[2] = (imid, cmid))."""
    return PretrainLossFn.apply(f, g, temperature, cmid_weight)


def ntxent_loss(z0, z1, temperature=0.1):
    return NTXentFn.apply(z0, z1, temperature)


_scale_aware_forwards(globals())
