"""torch_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

fp32 torch-CPU restatement of the floating-point part of the ViPFormer ``--mp``
pre-training hot path, written as pure functions over a state dict that uses
the reference's key names.  Index work (FPS / kNN / grouping) is delegated to
the C oracle (oracle/vpf_oracle.c) so that both halves are the same code the
golden fixtures pin.

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may
import this module; nothing under vipformer_amd/ does.

Parity status: PINNED for everything except NT-Xent.  tests/test_oracle_golden.py
checks these functions against fixtures captured from the imported reference
(tests/golden/make_golden.py).  ``ntxent`` restates lightly==1.1.21's
NTXentLoss (requirements.txt:4; call sites pretrain.py:155,196,202), which is
NOT in the reference tree and not installed here: that one function is
"parity unpinned" (restated from the published SimCLR formulation).

Reference line numbers are relative to the reference repo root.
"""
from __future__ import annotations

import ctypes
import math
import os
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def clib():
    """Load (building if needed) the C oracle."""
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libvpf_oracle.so")
        if not os.path.exists(so):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = ctypes.CDLL(so)
    return _LIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------- index stages
def fps_indices(pts: torch.Tensor, start_idx: torch.Tensor, G: int) -> torch.Tensor:
    """utils.py:56-85 with the random start (:71) supplied by the caller."""
    B, N, C = pts.shape
    p = np.ascontiguousarray(pts.detach().cpu().numpy(), dtype=np.float32)
    s = np.ascontiguousarray(start_idx.detach().cpu().numpy(), dtype=np.int64)
    out = np.zeros((B, G), dtype=np.int64)
    rc = clib().vpf_oracle_fps(_p(p), B, N, C, _p(s), G, _p(out))
    if rc != 0:
        raise RuntimeError(f"vpf_oracle_fps rc={rc}")
    return torch.from_numpy(out)


def square_distance(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """utils.py:122-141 (C == 3), bit-exact recipe."""
    B, Ns, _ = src.shape
    Nd = dst.shape[1]
    s = np.ascontiguousarray(src.numpy()[:, :, :3], dtype=np.float32)
    d = np.ascontiguousarray(dst.numpy()[:, :, :3], dtype=np.float32)
    out = np.zeros((B, Ns, Nd), dtype=np.float32)
    clib().vpf_oracle_square_distance(_p(s), _p(d), B, Ns, Nd, _p(out))
    return torch.from_numpy(out)


def knn_indices(K: int, pts: torch.Tensor, centers: torch.Tensor):
    """utils.py:107-119 in canonical order (ascending distance, ties -> lower index)."""
    B, N, C = pts.shape
    G = centers.shape[1]
    p = np.ascontiguousarray(pts.numpy(), dtype=np.float32)
    c = np.ascontiguousarray(centers.numpy(), dtype=np.float32)
    idx = np.zeros((B, G, K), dtype=np.int64)
    dist = np.zeros((B, G, K), dtype=np.float32)
    rc = clib().vpf_oracle_knn(_p(p), B, N, C, _p(c), G, K, _p(idx), _p(dist))
    if rc != 0:
        raise RuntimeError(f"vpf_oracle_knn rc={rc}")
    return torch.from_numpy(idx), torch.from_numpy(dist)


def divide_patches(pts: torch.Tensor, fps_idx: torch.Tensor, K: int, quirk: bool = True):
    """utils.py:6-38 given the FPS indices; returns (neighbors, centers, knn_idx)."""
    B, N, C = pts.shape
    G = fps_idx.shape[1]
    p = np.ascontiguousarray(pts.detach().numpy(), dtype=np.float32)
    f = np.ascontiguousarray(fps_idx.numpy(), dtype=np.int64)
    idx = np.zeros((B, G, K), dtype=np.int64)
    nb = np.zeros((B, G, K, C), dtype=np.float32)
    ct = np.zeros((B, G, C), dtype=np.float32)
    rc = clib().vpf_oracle_divide_patches(_p(p), B, N, C, _p(f), G, K, int(quirk), _p(idx), _p(nb), _p(ct))
    if rc != 0:
        raise RuntimeError(f"vpf_oracle_divide_patches rc={rc}")
    return torch.from_numpy(nb), torch.from_numpy(ct), torch.from_numpy(idx)


# --------------------------------------------------------------------------- config
@dataclass
class Arch:
    """The ~10 architecture flags of utils.py:119-149 (build_model, --mp branch)."""
    D: int = 256           # num_latent_channels
    H: int = 4             # num_ca_heads == num_sa_heads on every shipped script
    G: int = 96            # num_pc_latents
    K: int = 32            # group_size
    S: int = 6             # num_sa_layers
    n_ca: int = 1          # num_ca_layers
    MR: int = 2            # mlp_widen_factor
    atten_drop: float = 0.1
    mlp_drop: float = 0.5
    img: int = 224
    patch: int = 16
    point_channels: int = 3

    @property
    def T(self) -> int:
        return (self.img // self.patch) ** 2


class Masks:
    """Dropout keep-masks per site.  ``get(site, shape, p)`` returns a float keep mask
    (1/0) or None.  mode 'off' = eval; 'torch' = draw with torch's RNG in call order
    (what the reference does); 'given' = look the site up in ``table``."""

    def __init__(self, mode: str = "off", table: Optional[Dict[str, torch.Tensor]] = None):
        self.mode = mode
        self.table = table or {}
        self.drawn: Dict[str, torch.Tensor] = {}

    def apply(self, x: torch.Tensor, site: str, p: float) -> torch.Tensor:
        if self.mode == "off" or p == 0.0:
            return x
        if self.mode == "torch":
            return F.dropout(x, p, True)
        keep = self.table[site].to(x.dtype).reshape(x.shape)
        return x * keep * (1.0 / (1.0 - p))


# --------------------------------------------------------------------------- precision emulation
class _RoundBF16(torch.autograd.Function):
    """Round-to-nearest-even to bf16 and back, straight-through gradient."""

    @staticmethod
    def forward(ctx, x):
        return x.to(_FWD_DTYPE[0]).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGradBF16(torch.autograd.Function):
    """Identity forward; the gradient that arrives is rounded to bf16 -- the backward kernels store these gradients as the bf16
    operands of the next dgrad / wgrad product."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if _BWD_DTYPE[0] is torch.float16:        # loss-scaled fp16 gradient operands (the reference's GradScaler arithmetic, pretrain.py:154,209)
            return (g * _BWD_SCALE[0]).to(torch.float16).to(torch.float32) / _BWD_SCALE[0]
        return g.to(torch.bfloat16).to(torch.float32)


# Rounding points of the HIP path, by tag.  Forward (activations / weights stored as bf16):
#   w weights | adapter | g2e | pos | ln (LayerNorm outputs fed to a GEMM) | qkv | p (attention probabilities) | o (attention output)
#   | u (fc1 pre-activation) | h (GELU output) | head
# Backward (gradients stored as bf16 operands; only with emulate_bf16(backward=True)):
#   dz (gradient of a residual branch's output, behind dropout') | du | do | dqkv | ds (score gradient inside the attention backward)
FWD_TAGS = ("w", "adapter", "g2e", "pos", "ln", "qkv", "p", "o", "u", "h", "head")
BWD_TAGS = ("dz", "du", "do", "dqkv", "ds")
_EMULATE = [frozenset()]
_BWD_DTYPE = [torch.bfloat16]
_BWD_SCALE = [1.0]
_FWD_DTYPE = [torch.bfloat16]        # the type forward operands are rounded to (emulate_fp16 switches it; gradients stay bf16)


class emulate_bf16:
    """Context manager: while active, the float stages round to bf16 at exactly the points where the HIP
    path stores bf16 (GEMM operand weights, LayerNorm / activation / projection outputs, attention
    probabilities and outputs).  Off (the default) the oracle is the plain fp32 restatement that the
    golden fixtures pin against the reference.  On, it predicts the HIP path to ~1e-3 -- in particular
    the same max-pool winners -- which separates kernel logic errors from bf16 precision effects.

    only: an iterable of tags (FWD_TAGS / BWD_TAGS) -- round at those points alone (the per-stage error budget of
    tools/rounding_budget.py); backward=True: also round the GRADIENTS the backward kernels store as bf16."""

    def __init__(self, only=None, backward=False):
        if only is None:
            only = FWD_TAGS + (BWD_TAGS if backward else ())
        self.tags = frozenset(only)

    def __enter__(self):
        self.prev = _EMULATE[0]
        _EMULATE[0] = self.tags

    def __exit__(self, *a):
        _EMULATE[0] = self.prev


class emulate_fp16(emulate_bf16):
    """The same rounding points with the FORWARD operands (weights + activations) rounded to IEEE fp16 -- the reference's own
    autocast dtype (pretrain.py:154,176): 11 significant bits instead of bf16's 8, range 6e-8 .. 65504 (a value beyond it becomes
    inf exactly as `v_cvt_f16_f32` would make it).  The gradients the backward kernels store stay bf16 (range).  Used by
    tests/rounding_budget.py to measure what fp16 forward operands would buy against SURVEY 8c's tolerances."""

    def __init__(self, only=None, backward=False, grad_scale=None):
        """grad_scale: None -- gradient operands stay bf16; a number -- they are fp16 as well, rounded at ``grad_scale`` times their
        value (GradScaler's loss scale: the gradients of the whole backward pass carry that factor)."""
        super().__init__(only, backward)
        self.grad_scale = grad_scale

    def __enter__(self):
        super().__enter__()
        self.prev_dt = (_FWD_DTYPE[0], _BWD_DTYPE[0], _BWD_SCALE[0])
        _FWD_DTYPE[0] = torch.float16
        if self.grad_scale is not None:
            _BWD_DTYPE[0], _BWD_SCALE[0] = torch.float16, float(self.grad_scale)

    def __exit__(self, *a):
        _FWD_DTYPE[0], _BWD_DTYPE[0], _BWD_SCALE[0] = self.prev_dt
        super().__exit__(*a)


def Q(x, tag="w"):
    return _RoundBF16.apply(x) if tag in _EMULATE[0] else x


def QB(x, tag):
    return _RoundGradBF16.apply(x) if tag in _EMULATE[0] else x


# --------------------------------------------------------------------------- float stages
def adapter(sd, pre: str, pts):
    """classifier.py:31-36,48: Linear(C,64) -> LayerNorm(64) -> ReLU -> Linear(64,D)."""
    h = F.linear(pts, sd[pre + "point_mlp.0.weight"], sd[pre + "point_mlp.0.bias"])
    h = F.layer_norm(h, (64,), sd[pre + "point_mlp.1.weight"], sd[pre + "point_mlp.1.bias"], 1e-5)
    h = Q(F.relu(h), "adapter")
    return Q(F.linear(h, Q(sd[pre + "point_mlp.3.weight"]), sd[pre + "point_mlp.3.bias"]), "adapter")


def _bn(sd, pre, x, train, buffers):
    """BatchNorm1d as utils.py:156,163 / partseg.py:520,523 (momentum .1, eps 1e-5).
    ``buffers`` (dict) receives the updated running stats when train."""
    rm, rv = sd[pre + "running_mean"], sd[pre + "running_var"]
    if train:
        rm, rv = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm, rv, sd[pre + "weight"], sd[pre + "bias"], train, 0.1, 1e-5)
    if train and buffers is not None:
        buffers[pre + "running_mean"] = rm
        buffers[pre + "running_var"] = rv
        buffers[pre + "num_batches_tracked"] = sd[pre + "num_batches_tracked"] + 1
    return y


def group2emb(sd, pre: str, neighbors, train: bool, buffers=None):
    """utils.py:168-189.  neighbors [B,G,K,C] -> [B,G,D]."""
    B, G, K, C = neighbors.shape
    x = neighbors.reshape(B * G, K, C).transpose(2, 1)                       # [BG, C, K]
    h = F.conv1d(x, sd[pre + "first_conv.0.weight"], sd[pre + "first_conv.0.bias"])
    h = Q(F.relu(_bn(sd, pre + "first_conv.1.", h, train, buffers)), "g2e")
    h = Q(F.conv1d(h, Q(sd[pre + "first_conv.3.weight"]), sd[pre + "first_conv.3.bias"]), "g2e")   # [BG,128,K]
    g = h.max(dim=2, keepdim=True)[0]
    h = torch.cat([g.expand(-1, -1, K), h], dim=1)                            # [BG,256,K]
    h = Q(F.conv1d(h, Q(sd[pre + "second_conv.0.weight"]), sd[pre + "second_conv.0.bias"]), "g2e")
    h = Q(F.relu(_bn(sd, pre + "second_conv.1.", h, train, buffers)), "g2e")
    h = Q(F.conv1d(h, Q(sd[pre + "second_conv.3.weight"]), sd[pre + "second_conv.3.bias"]), "g2e")
    return h.max(dim=2)[0].reshape(B, G, -1)


def pos_mlp(sd, pre: str, centers):
    """partseg.py:498-501: Linear(3,128) -> GELU(erf) -> Linear(128,D)."""
    h = Q(F.gelu(F.linear(centers, sd[pre + "0.weight"], sd[pre + "0.bias"])), "pos")
    return F.linear(h, Q(sd[pre + "2.weight"]), sd[pre + "2.bias"])


def mha(sd, pre: str, xq, xkv, H: int, p: float, masks: Masks, site: str, pad_mask=None):
    """partseg.py:53-86.  pad_mask: bool [B, Lkv], True = padding key (:73-77)."""
    B, Lq, D = xq.shape
    Lk = xkv.shape[1]
    dh = D // H
    xq, xkv = Q(xq, "ln"), Q(xkv, "ln")
    q = QB(Q(F.linear(xq, Q(sd[pre + "q_proj.weight"])), "qkv"), "dqkv").reshape(B, Lq, H, dh).permute(0, 2, 1, 3).reshape(B * H, Lq, dh)
    k = QB(Q(F.linear(xkv, Q(sd[pre + "k_proj.weight"])), "qkv"), "dqkv").reshape(B, Lk, H, dh).permute(0, 2, 1, 3).reshape(B * H, Lk, dh)
    v = QB(Q(F.linear(xkv, Q(sd[pre + "v_proj.weight"])), "qkv"), "dqkv").reshape(B, Lk, H, dh).permute(0, 2, 1, 3).reshape(B * H, Lk, dh)
    a = QB(torch.bmm(q, k.transpose(1, 2)), "ds") * (dh ** -0.5)
    if pad_mask is not None:
        pm = pad_mask.bool()[:, None, None, :].expand(B, H, 1, Lk).reshape(B * H, 1, Lk)      # "b j -> (b h) () j"
        a = a.masked_fill(pm, -torch.finfo(a.dtype).max)
    a = a.softmax(dim=-1)
    a = Q(masks.apply(a, site, p), "p")
    o = QB(Q(torch.bmm(a, v), "o"), "do").reshape(B, H, Lq, dh).permute(0, 2, 1, 3).reshape(B, Lq, D)
    return QB(F.linear(o, Q(sd[pre + "o_proj.weight"]), sd[pre + "o_proj.bias"]), "dz")


def mlp(sd, pre: str, x):
    """partseg.py:191-198: LN -> Linear -> GELU -> Linear."""
    D = x.shape[-1]
    h = Q(F.layer_norm(x, (D,), sd[pre + "0.weight"], sd[pre + "0.bias"], 1e-5), "ln")
    h = Q(F.gelu(QB(Q(F.linear(h, Q(sd[pre + "1.weight"]), sd[pre + "1.bias"]), "u"), "du")), "h")
    return QB(F.linear(h, Q(sd[pre + "3.weight"]), sd[pre + "3.bias"]), "dz")


def ca_layer(sd, pre: str, xq, xkv, a: Arch, masks: Masks, tag: str, pad_mask=None):
    """partseg.py:144-167 + Residual :201-213.  Residual dropout p = atten_drop for the
    attention branch (:165) and mlp_drop for the MLP branch (:166)."""
    D = xq.shape[-1]
    m = pre + "0.module."
    nq = F.layer_norm(xq, (D,), sd[m + "q_norm.weight"], sd[m + "q_norm.bias"], 1e-5)
    nk = F.layer_norm(xkv, (D,), sd[m + "kv_norm.weight"], sd[m + "kv_norm.bias"], 1e-5)
    y = mha(sd, m + "attention.", nq, nk, a.H, a.atten_drop, masks, tag + ".attn", pad_mask)
    x = masks.apply(y, tag + ".res1", a.atten_drop) + xq
    y = mlp(sd, pre + "1.module.", x)
    return masks.apply(y, tag + ".res2", a.mlp_drop) + x


def sa_layer(sd, pre: str, x, a: Arch, masks: Masks, tag: str):
    """partseg.py:170-188.  BOTH residual dropouts use mlp_drop (:186-187)."""
    D = x.shape[-1]
    m = pre + "0.module."
    n = F.layer_norm(x, (D,), sd[m + "norm.weight"], sd[m + "norm.bias"], 1e-5)
    y = mha(sd, m + "attention.", n, n, a.H, a.atten_drop, masks, tag + ".attn")
    x1 = masks.apply(y, tag + ".res1", a.mlp_drop) + x
    y = mlp(sd, pre + "1.module.", x1)
    return masks.apply(y, tag + ".res2", a.mlp_drop) + x1


def encoder(sd, pre: str, tokens, pos, kv, a: Arch, masks: Masks, taps=(), pad_mask=None):
    """partseg.py:314-342.  pos is re-added before every layer and is part of the
    residual base; pad_mask reaches the cross-attention layers only (:326,334).  Returns (x, [tapped layer outputs])."""
    x = ca_layer(sd, pre + "cross_attn_1.", tokens + pos, kv, a, masks, "ca", pad_mask)
    feats = []
    for i in range(a.S):
        if i + 1 < a.n_ca:
            x = ca_layer(sd, pre + "cross_attn_n.", x + pos, kv, a, masks, f"ca{i + 1}", pad_mask)
        x = sa_layer(sd, pre + f"sa_layers.{i}.", x + pos, a, masks, f"sa{i}")
        if (i + 1) in taps:
            feats.append(x)
    return x, feats


def latent_head(sd, pre: str, x, train: bool, buffers=None):
    """partseg.py:519-525: BN1d(2D) ReLU Linear(2D,D,no bias) BN1d(D) ReLU Linear(D,D,no bias)."""
    h = Q(F.relu(_bn(sd, pre + "0.", x, train, buffers)), "head")
    h = F.linear(h, Q(sd[pre + "2.weight"]))
    h = Q(F.relu(_bn(sd, pre + "3.", h, train, buffers)), "head")
    return F.linear(h, Q(sd[pre + "5.weight"]))


def pool(x):
    """partseg.py:547: cat[max over tokens, mean over tokens]."""
    return torch.cat([x.max(1)[0], x.mean(1)], dim=1)


def pc_forward(sd, pts, start_idx, a: Arch, train: bool, masks: Optional[Masks] = None, buffers=None):
    """CrossFormer_pc_mp.forward, partseg.py:527-550.  Returns (feats, backbone)."""
    masks = masks or Masks("off")
    kv = adapter(sd, "input_adapter.", pts)
    fidx = fps_indices(pts, start_idx, a.G)
    nb, ct, _ = divide_patches(pts, fidx, a.K, True)
    tok = group2emb(sd, "group2emb.", nb, train, buffers)
    pos = pos_mlp(sd, "position_emb.", ct)
    x, _ = encoder(sd, "encoder.", tok, pos, kv, a, masks)
    bb = pool(x)
    return latent_head(sd, "latent_head.", bb, train, buffers), bb


def finetune_head(sd, pre: str, x, train: bool, buffers=None):
    """partseg.py:572-581: three blocks BN1d - ReLU - Linear (with bias): 2D -> D -> D/2 -> num_obj_classes."""
    h = x
    for i in (0, 3, 6):
        h = Q(F.relu(_bn(sd, pre + f"{i}.", h, train, buffers)))
        h = F.linear(h, Q(sd[pre + f"{i + 2}.weight"]), sd[pre + f"{i + 2}.bias"])
    return h


def pc_ft_forward(sd, pts, start_idx, a: Arch, train: bool, masks: Optional[Masks] = None, buffers=None):
    """CrossFormer_pc_mp_ft.forward, partseg.py:583-605 (the ModelNet fine-tuning classifier): the pre-training backbone
    (pc_forward above without latent_head) followed by finetune_head.  Returns the logits [B, num_obj_classes]."""
    masks = masks or Masks("off")
    kv = adapter(sd, "input_adapter.", pts)
    fidx = fps_indices(pts, start_idx, a.G)
    nb, ct, _ = divide_patches(pts, fidx, a.K, True)
    tok = group2emb(sd, "group2emb.", nb, train, buffers)
    pos = pos_mlp(sd, "position_emb.", ct)
    x, _ = encoder(sd, "encoder.", tok, pos, kv, a, masks)
    return finetune_head(sd, "finetune_head.", pool(x), train, buffers)


def three_nn(xyz1, xyz2):
    """utils.py:219-230: for every point of xyz1 [B,N,3] the three nearest of xyz2 [B,S,3] by square_distance (bit-exact recipe,
    C oracle) and their normalised inverse-distance weights; canonical tie order (stable sort = lower index first)."""
    d = square_distance(xyz1.detach(), xyz2.detach())
    dist, idx = torch.sort(d, dim=-1, stable=True)
    dist, idx = dist[:, :, :3], idx[:, :, :3]
    recip = 1.0 / (dist + 1e-8)
    return idx, recip / recip.sum(dim=2, keepdim=True)


def feature_propagation(sd, pre: str, xyz1, xyz2, points1, points2, train: bool, buffers=None):
    """PointNetFeaturePropagation.forward, utils.py:205-242, in row-major layouts: xyz1 [B,N,3], xyz2 [B,S,3], points1 [B,N,C1] or
    None, points2 [B,S,F] -> [B,N,mlp[-1]]."""
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    if S == 1:
        interp = points2.repeat(1, N, 1)
    else:
        idx, w = three_nn(xyz1, xyz2)
        gathered = torch.gather(points2.unsqueeze(1).expand(B, N, S, points2.shape[2]), 2,
                                idx.unsqueeze(-1).expand(B, N, 3, points2.shape[2]))
        interp = torch.sum(gathered * w.view(B, N, 3, 1), dim=2)
    x = torch.cat([points1, interp], dim=-1) if points1 is not None else interp
    x = Q(x).permute(0, 2, 1)
    i = 0
    while pre + f"mlp_convs.{i}.weight" in sd:
        h = F.conv1d(x, Q(sd[pre + f"mlp_convs.{i}.weight"]), sd[pre + f"mlp_convs.{i}.bias"])      # (pre-BatchNorm: fp32 on the HIP path)
        x = Q(F.relu(_bn(sd, pre + f"mlp_bns.{i}.", h, train, buffers)))
        i += 1
    return x.permute(0, 2, 1)


def partseg_forward(sd, pts, start_idx, cls_label, a: Arch, layer_idx, train: bool, masks: Optional[Masks] = None, buffers=None,
                    head_mask=None):
    """CrossFormer_partseg.forward, partseg.py:407-470 -> logits [B,N,num_part_classes].  head_mask: keep mask [B*N,512] of dp1
    (None: torch's RNG when train, nothing in eval)."""
    masks = masks or Masks("off")
    B, N, _ = pts.shape
    kv = adapter(sd, "input_adapter.", pts)
    fidx = fps_indices(pts, start_idx, a.G)
    nb, ct, _ = divide_patches(pts, fidx, a.K, True)
    tok = group2emb(sd, "group2emb.", nb, train, buffers)
    pos = pos_mlp(sd, "position_emb.", ct)
    _, feats = encoder(sd, "encoder.", tok, pos, kv, a, masks, taps=tuple(layer_idx))
    D = a.D
    x = torch.cat([F.layer_norm(f, (D,), sd["norm.weight"], sd["norm.bias"], 1e-5) for f in feats], dim=2)      # [B,G,nl*D]
    x_max, x_avg = x.max(dim=1)[0], x.mean(dim=1)
    lab = F.conv1d(Q(cls_label.view(B, 16, 1)), Q(sd["label_conv.0.weight"]))
    lab = Q(F.leaky_relu(_bn(sd, "label_conv.1.", lab, train, buffers), 0.2)).view(B, 64)
    glob = Q(torch.cat([x_max, x_avg, lab], dim=1))                                                             # [B, 2*nl*D + 64]
    f0 = feature_propagation(sd, "propagation.", pts[:, :, :3], ct[:, :, :3], pts, x, train, buffers)          # [B,N,1024]
    h = torch.cat([f0, glob.unsqueeze(1).expand(B, N, glob.shape[1])], dim=2).permute(0, 2, 1)                  # [B, 1024 + ..., N]
    h = F.conv1d(h, Q(sd["conv1.weight"]), sd["conv1.bias"])
    h = Q(F.relu(_bn(sd, "bn1.", h, train, buffers)))
    if train:
        if head_mask is not None:
            h = h * head_mask.view(B, N, -1).permute(0, 2, 1).to(h.dtype) * 2.0
        else:
            h = F.dropout(h, 0.5, True)
        h = Q(h)
    h = F.conv1d(h, Q(sd["conv2.weight"]), sd["conv2.bias"])
    h = Q(F.relu(_bn(sd, "bn2.", h, train, buffers)))
    h = F.conv1d(h, Q(sd["conv3.weight"]), sd["conv3.bias"])
    return h.permute(0, 2, 1)


def patchify(imgs, p: int):
    """partseg.py:632: 'b (h p1) (w p2) c -> b (h w) (p1 p2 c)'."""
    B, Hh, Ww, C = imgs.shape
    x = imgs.reshape(B, Hh // p, p, Ww // p, p, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B, (Hh // p) * (Ww // p), p * p * C)


def img_forward(sd, imgs, a: Arch, train: bool, masks: Optional[Masks] = None, buffers=None):
    """CrossFormer_img_mp.forward, partseg.py:661-680.  imgs [B,H,W,3]."""
    masks = masks or Masks("off")
    tok = F.linear(Q(patchify(imgs, a.patch)), Q(sd["patch2emb.1.weight"]), sd["patch2emb.1.bias"])
    x, _ = encoder(sd, "encoder.", tok, sd["position_emb"], tok, a, masks)
    bb = pool(x)
    return latent_head(sd, "latent_head.", bb, train, buffers), bb


def ntxent(z0, z1, temperature: float = 0.1):
    """lightly==1.1.21 NTXentLoss(memory_bank_size=0) -- PARITY UNPINNED (third party,
    absent from the reference tree; call sites pretrain.py:155,196,202).  Restates
    lightly/loss/ntx_ent_loss.py::NTXentLoss.forward as published at tag v1.1.21 (the
    `memory_bank_size == 0` branch: no negatives from a bank, and -- in that version -- no
    gather across ranks):
    L2-normalise; out = cat(z0,z1); logits = out.out^T / T with the diagonal removed;
    positive of row i is row i+b (mod 2b); mean cross-entropy over the 2b rows."""
    b = z0.shape[0]
    z = torch.cat([F.normalize(z0, dim=1), F.normalize(z1, dim=1)], 0)
    logits = z @ z.t() / temperature
    n = 2 * b
    eye = torch.eye(n, dtype=torch.bool)
    logits = logits[~eye].view(n, n - 1)
    labels = torch.cat([torch.arange(b) + b - 1, torch.arange(b)])
    return F.cross_entropy(logits, labels)


def pretrain_losses(pc_sd, img_sd, pc_t1, pc_t2, imgs, start_idx, a: Arch, train=True,
                    pc_masks=None, img_masks=None, pc_buffers=None, img_buffers=None,
                    cmid_weight: float = 1.0):
    """pretrain.py:183-207 (modality 'both').  imgs is [b,H,W,3]."""
    b = pc_t1.shape[0]
    pc = torch.cat([pc_t1, pc_t2], 0)
    f, _ = pc_forward(pc_sd, pc, start_idx, a, train, pc_masks, pc_buffers)
    f1, f2 = f[:b], f[b:]
    l_im = ntxent(f1, f2)
    fi, _ = img_forward(img_sd, imgs, a, train, img_masks, img_buffers)
    l_cm = ntxent((f1 + f2) / 2, fi)
    return l_im + cmid_weight * l_cm, l_im, l_cm


def adamw_step(params: Dict[str, torch.Tensor], grads, state, step: int, lr=1e-3, betas=(0.9, 0.999),
               eps=1e-8, wd=0.01):
    """torch.optim.AdamW defaults as used by pretrain.py:121-124 (lr 1e-3, wd 0.01)."""
    b1, b2 = betas
    for k, p in params.items():
        g = grads[k]
        m, v = state.setdefault(k, (torch.zeros_like(p), torch.zeros_like(p)))
        p.mul_(1 - lr * wd)
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)
