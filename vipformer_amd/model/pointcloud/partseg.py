"""MI355X mirror of the pre-training part of ``vipformer.model.pointcloud.partseg``
(reference partseg.py:14-342, 473-550, 608-680): same class names, constructor
arguments, forward contracts and state-dict keys; every forward is a short sequence
of C-ABI calls into libvipformer_hip.so (see vipformer_amd/ops.py).

Supported configuration of the fused kernels: head dim 64 (every shipped architecture:
H4D256, H6D384), q/kv/latent/output channels all equal, no pad/attn masks (never used on
the pre-training path, partseg.py:326-335 passes pad_mask=None).  Anything else raises.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from ... import _lib as L
from ... import ops
from .utils import Group2Emb, Sequential, divide_patches


def _ln_params(*mods):
    out = []
    for m in mods:
        if m is not None:
            out += list(m.parameters())
    return out


class MultiHeadAttention(nn.Module):
    """partseg.py:14-86.  q/k/v projections without bias, o_proj with bias, dropout on the
    attention probabilities.  Returns fp32 [B, Lq, D]."""

    def __init__(self, num_heads: int, num_q_input_channels: int, num_kv_input_channels: int, num_latent_channels: int,
                 num_output_channels: Optional[int] = None, dropout: float = 0.0):
        super().__init__()
        if num_output_channels is None:
            num_output_channels = num_q_input_channels
        if num_latent_channels % num_heads != 0:
            raise ValueError("num_latent_channels must be divisible by num_heads")
        per_head = num_latent_channels // num_heads
        self.dp_scale = per_head ** -0.5
        self.num_heads = num_heads
        self.q_proj = nn.Linear(num_q_input_channels, num_latent_channels, bias=False)
        self.k_proj = nn.Linear(num_kv_input_channels, num_latent_channels, bias=False)
        self.v_proj = nn.Linear(num_kv_input_channels, num_latent_channels, bias=False)
        self.o_proj = nn.Linear(num_latent_channels, num_latent_channels)
        self.dropout = nn.Dropout(dropout)
        self.site_attn = ops.new_site()
        self._dims = (num_q_input_channels, num_kv_input_channels, num_latent_channels, num_output_channels, per_head)

    def _check(self, pad_mask, attn_mask):
        if attn_mask is not None:
            raise NotImplementedError("attention masks not supported yet")
        qi, ki, lat, out, ph = self._dims
        if not (qi == ki == lat == out) or ph != 64:
            raise L.VpfError("fused attention needs equal q/kv/latent/output channels and head dim 64")

    def _run(self, x_q, x_kv, pos, ln_q, ln_kv, residual, p_res, site_res, pad_mask=None):
        cfg = dict(ln_q=ln_q, ln_kv=ln_kv, residual=residual, p_res=p_res, site_res=site_res, training=self.training)
        if pad_mask is not None:
            # partseg.py:54,73-76: boolean [B, Lkv], True = padding key; the kernels read it as bytes (vpf_attention_fwd_pad)
            lkv = (x_q if x_kv is None else x_kv).shape[1]
            if pad_mask.dim() != 2 or tuple(pad_mask.shape) != (x_q.shape[0], lkv):
                raise L.VpfError(f"pad_mask must be [B, Lkv] = {(x_q.shape[0], lkv)}, got {tuple(pad_mask.shape)}")
            cfg["pad_mask"] = (pad_mask != 0).to(torch.uint8).contiguous()
        return ops.AttnBlockFn.apply(x_q, pos, x_kv, self, cfg, *(list(self.parameters()) + _ln_params(ln_q, ln_kv)))

    def forward(self, x_q, x_kv, pad_mask=None, attn_mask=None):
        self._check(pad_mask, attn_mask)
        return self._run(x_q, None if x_kv is x_q else x_kv, None, None, None, False, 0.0, 0, pad_mask)


class CrossAttention(nn.Module):
    """partseg.py:89-116: LayerNorm(q), LayerNorm(kv) (separate affine) then MultiHeadAttention."""

    def __init__(self, num_heads: int, num_q_input_channels: int, num_kv_input_channels: int, num_latent_channels: int,
                 dropout: float = 0.0):
        super().__init__()
        self.q_norm = nn.LayerNorm(num_q_input_channels)
        self.kv_norm = nn.LayerNorm(num_kv_input_channels)
        self.attention = MultiHeadAttention(num_heads=num_heads, num_q_input_channels=num_q_input_channels,
                                            num_kv_input_channels=num_kv_input_channels,
                                            num_latent_channels=num_latent_channels,
                                            num_output_channels=num_latent_channels, dropout=dropout)

    def _fused(self, args, pos, residual, p_res, site_res):
        x_q, x_kv = args[0], args[1]
        pad_mask = args[2] if len(args) > 2 else None
        self.attention._check(pad_mask, None)
        return self.attention._run(x_q, x_kv, pos, self.q_norm, self.kv_norm, residual, p_res, site_res, pad_mask)

    def forward(self, x_q, x_kv, pad_mask=None, attn_mask=None):
        self.attention._check(pad_mask, attn_mask)
        return self.attention._run(x_q, x_kv, None, self.q_norm, self.kv_norm, False, 0.0, 0, pad_mask)


class SelfAttention(nn.Module):
    """partseg.py:119-141: one LayerNorm then MultiHeadAttention(x, x)."""

    def __init__(self, num_heads: int, num_latent_channels: int, dropout: float = 0.0):
        super().__init__()
        self.norm = nn.LayerNorm(num_latent_channels)
        self.attention = MultiHeadAttention(num_heads=num_heads, num_q_input_channels=num_latent_channels,
                                            num_kv_input_channels=num_latent_channels,
                                            num_latent_channels=num_latent_channels,
                                            num_output_channels=num_latent_channels, dropout=dropout)

    def _fused(self, args, pos, residual, p_res, site_res):
        pad_mask = args[1] if len(args) > 1 else None
        self.attention._check(pad_mask, None)
        return self.attention._run(args[0], None, pos, self.norm, None, residual, p_res, site_res, pad_mask)

    def forward(self, x, pad_mask=None, attn_mask=None):
        self.attention._check(pad_mask, attn_mask)
        return self.attention._run(x, None, None, self.norm, None, False, 0.0, 0, pad_mask)


class MLP(Sequential):
    """partseg.py:191-198: LayerNorm -> Linear(D, w*D) -> GELU(erf) -> Linear(w*D, D)."""

    def __init__(self, num_channels: int, widening_factor: int):
        super().__init__(nn.LayerNorm(num_channels), nn.Linear(num_channels, widening_factor * num_channels), nn.GELU(),
                         nn.Linear(widening_factor * num_channels, num_channels))

    def _fused(self, args, pos, residual, p_res, site_res):
        if pos is not None:
            raise L.VpfError("pos is only fused into attention blocks")
        cfg = dict(residual=residual, p_res=p_res, site_res=site_res, training=self.training)
        return ops.MLPBlockFn.apply(args[0], self, cfg, *self.parameters())

    def forward(self, *x):
        return self._fused(x, None, False, 0.0, 0)


class DropPath(nn.Module):
    """Stochastic depth (timm.models.layers.DropPath in the reference).  Identity on the
    pre-training path (max_dpr = 0 on every shipped script, partseg.py:206)."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask / keep


class Residual(nn.Module):
    """partseg.py:201-213: drop_path(dropout(module(*args)) + args[0]).  For the attention and
    MLP blocks the dropout + residual add run inside the producing GEMM's epilogue."""

    def __init__(self, module: nn.Module, dropout: float, drop_path_rate: float):
        super().__init__()
        self.module = module
        self.dropout = nn.Dropout(p=dropout)
        self.drop_path = DropPath(drop_path_rate) if drop_path_rate > .0 else nn.Identity()
        self.site = ops.new_site()

    def forward(self, *args, pos=None, **kwargs):
        fused = isinstance(self.module, (CrossAttention, SelfAttention, MLP)) and not kwargs \
            and isinstance(self.drop_path, nn.Identity)
        if fused:
            return self.module._fused(args, pos, True, self.dropout.p, self.site)
        base = args[0] if pos is None else args[0] + pos
        y = self.module(base, *args[1:], **kwargs)
        p = self.dropout.p if self.training else 0.0
        return self.drop_path(ops.DropoutAddFn.apply(y, base, p, self.site))


class _Layer(Sequential):
    """Two Residual blocks; ``pos`` (added to the input, part of the residual base,
    partseg.py:326,335) is folded into the first block's LayerNorm kernel."""

    def forward(self, *x, pos=None):
        x = self[0](*x, pos=pos)
        return self[1](x)


class CrossAttentionLayer(_Layer):
    """partseg.py:144-167.  Residual dropout: attention branch atten_drop, MLP branch mlp_drop."""

    def __init__(self, num_heads: int, num_q_input_channels: int, num_kv_input_channels: int, num_latent_channels: int,
                 widening_factor: int = 1, drop_path_rate: float = 0.0, atten_drop: float = 0.0, mlp_drop: float = 0.0,
                 attention_residual: bool = True):
        cross_attn = CrossAttention(num_heads=num_heads, num_q_input_channels=num_q_input_channels,
                                    num_kv_input_channels=num_kv_input_channels,
                                    num_latent_channels=num_latent_channels, dropout=atten_drop)
        self.attention_residual = attention_residual
        super().__init__(Residual(cross_attn, atten_drop, drop_path_rate) if attention_residual else cross_attn,
                         Residual(MLP(num_q_input_channels, widening_factor), mlp_drop, drop_path_rate))

    def forward(self, *x, pos=None):
        if not self.attention_residual:
            base = x[0] if pos is None else x[0] + pos
            return self[1](self[0](base, *x[1:]))
        return super().forward(*x, pos=pos)


class SelfAttentionLayer(_Layer):
    """partseg.py:170-188.  BOTH residual dropouts use mlp_drop (:186-187)."""

    def __init__(self, num_heads: int, num_latent_channels: int, widening_factor: int = 1, drop_path_rate: float = 0.0,
                 atten_drop: float = 0.0, mlp_drop: float = 0.0):
        self_attn = SelfAttention(num_heads=num_heads, num_latent_channels=num_latent_channels, dropout=atten_drop)
        super().__init__(Residual(self_attn, mlp_drop, drop_path_rate),
                         Residual(MLP(num_latent_channels, widening_factor), mlp_drop, drop_path_rate))


class Encoder(nn.Module):
    """partseg.py:233-342: one cross-attention layer (query = tokens + pos, key/value = the
    per-point / per-patch embeddings) followed by ``num_self_attention_layers`` self-attention
    layers; ``pos`` is re-added before every layer.  ``cross_attn_1 is cross_attn_n`` when
    there is a single cross-attention layer (one parameter set, two state-dict prefixes)."""

    def __init__(self, num_latent_channels: int, num_cross_attention_layers: int = 1, num_cross_attention_heads: int = 4,
                 cross_attention_widening_factor: int = 1, first_cross_attention_layer_shared: bool = False,
                 num_self_attention_layers: int = 6, num_self_attention_heads: int = 4,
                 self_attention_widening_factor: int = 1, dpr_list: list = [], atten_drop: float = 0.0,
                 mlp_drop: float = 0.0, activation_checkpointing: bool = False, modal_prior: bool = False):
        super().__init__()
        if num_cross_attention_layers <= 0:
            raise ValueError("num_cross_attention_layers must be > 0")
        if activation_checkpointing:
            raise L.VpfError("activation checkpointing is never enabled by the reference's factories (utils.py:120-132)")
        self.num_cross_attention_layers = num_cross_attention_layers

        def cross_attn():
            return CrossAttentionLayer(num_heads=num_cross_attention_heads, num_q_input_channels=num_latent_channels,
                                       num_kv_input_channels=num_latent_channels, num_latent_channels=num_latent_channels,
                                       widening_factor=cross_attention_widening_factor, atten_drop=atten_drop,
                                       mlp_drop=mlp_drop)

        self.cross_attn_n = cross_attn()
        if first_cross_attention_layer_shared or num_cross_attention_layers == 1:
            self.cross_attn_1 = self.cross_attn_n
        else:
            self.cross_attn_1 = cross_attn()
        self.sa_layers = nn.ModuleList()
        for i in range(num_self_attention_layers):
            self.sa_layers.append(SelfAttentionLayer(num_heads=num_self_attention_heads,
                                                     num_latent_channels=num_latent_channels,
                                                     widening_factor=self_attention_widening_factor,
                                                     drop_path_rate=dpr_list[i], atten_drop=atten_drop,
                                                     mlp_drop=mlp_drop))
        self.modal_prior = modal_prior

    def fused_ok(self, group_embs, pts_embs, layer_idx=(), pad_mask=None):
        return (self.num_cross_attention_layers == 1 and pad_mask is None
                and ops.encoder_fused_supported(self.cross_attn_1, self.sa_layers, group_embs, pts_embs))

    def forward(self, group_embs, pos_embs, pts_embs, layer_idx=[], pad_mask=None, kv_ready=False):
        """kv_ready (not in the reference signature, default off): pts_embs already holds the cross-attention K | V
        projections produced by ops.AdapterKVFn."""
        feats = []
        if self.fused_ok(group_embs, pts_embs, layer_idx, pad_mask):
            # fused row-block kernels for the cross-attention layer's tail and every self-attention layer
            mods = [self.cross_attn_1] + list(self.sa_layers)
            params = [p for m in mods for p in m.parameters()]
            taps = tuple(sorted({int(i) for i in layer_idx if 1 <= int(i) <= len(self.sa_layers)}))      # partseg.py:336: i+1 in layer_idx
            outs = ops.EncoderFusedFn.apply(group_embs, pos_embs, pts_embs, self.cross_attn_1, self.sa_layers, self.training, kv_ready,
                                            taps, *params)
            return outs[0] if self.modal_prior else list(outs[1:])
        self.cross_attn_1.__dict__.pop("_vpf_front_stash", None)        # a CaFrontFn stash is for the fused path only: never left behind
        if kv_ready:
            raise L.VpfError("kv_ready needs the fused encoder path")
        x = self.cross_attn_1(group_embs, pts_embs, pad_mask, pos=pos_embs)
        if (self.num_cross_attention_layers == 1 and not layer_idx and pad_mask is None
                and ops.sa_stack_supported(self.sa_layers, x)):
            # one fused kernel per self-attention layer (vpf_sa_layer_fwd)
            params = [p for sa in self.sa_layers for p in sa.parameters()]
            x = ops.SAStackFn.apply(x, pos_embs, self.sa_layers, self.training, *params)
            return x if self.modal_prior else feats
        for i, sa in enumerate(self.sa_layers):
            if i + 1 < self.num_cross_attention_layers:
                x = self.cross_attn_n(x, pts_embs, pad_mask, pos=pos_embs)
            x = sa(x, pos=pos_embs)
            if i + 1 in layer_idx:
                feats.append(x)
        return x if self.modal_prior else feats


class _Patchify(nn.Module):
    """Stands in for einops' Rearrange('b (h p1) (w p2) c -> b (h w) (p1 p2 c)') at index 0 of
    ``patch2emb`` (keeps the Linear at ``patch2emb.1``); the gather itself is vpf_patchify."""

    def __init__(self, patch_size: int):
        super().__init__()
        self.patch_size = patch_size

    def forward(self, x):
        raise L.VpfError("patch2emb is executed as one fused op by CrossFormer_img_mp.forward")


def _latent_head(D):
    return nn.Sequential(nn.BatchNorm1d(2 * D), nn.ReLU(), nn.Linear(2 * D, D, bias=False), nn.BatchNorm1d(D), nn.ReLU(),
                         nn.Linear(D, D, bias=False))


def _encoder(D, n_ca, h_ca, n_sa, h_sa, mr, max_dpr, atten_drop, mlp_drop, modal_prior):
    dpr_list = [x.item() for x in torch.linspace(0, max_dpr, n_sa)]
    return Encoder(num_latent_channels=D, num_cross_attention_layers=n_ca, num_cross_attention_heads=h_ca,
                   cross_attention_widening_factor=mr, num_self_attention_layers=n_sa, num_self_attention_heads=h_sa,
                   self_attention_widening_factor=mr, dpr_list=dpr_list, atten_drop=atten_drop, mlp_drop=mlp_drop,
                   modal_prior=modal_prior)


class _PointBackbone(nn.Module):
    """What CrossFormer_pc_mp, CrossFormer_pc_mp_ft and CrossFormer_partseg share (partseg.py:407-425, 527-545): per-point adapter,
    divide_patches, Group2Emb, position MLP, Encoder.  Sub-classes create the members under the reference's names."""

    def backbone(self, pts, _groups=None, _cut=None, _after_g2e=None):
        return ops.PoolFn.apply(self._encode(pts, (), _groups, _cut, _after_g2e)[0])

    def _cut_here(self, _cut, *tensors):
        """_cut (a list, trainer extension): the encoder consumes DETACHED copies of its inputs and the (produced, consumed) pairs are
        recorded, so that backward can run in two calls -- everything down to the encoder's inputs first, then (from the detached
        copies' .grad) the input stages: Group2Emb, position MLP, point adapter / K,V producer.  Between the two a data-parallel
        trainer already sends the gradients the first call completed (train.Pretrainer)."""
        if _cut is None:
            return tensors
        out = []
        for t in tensors:
            d = t.detach().requires_grad_(t.requires_grad)
            if t.requires_grad:
                _cut.append((t, d))
            out.append(d)
        return tuple(out)

    def _encode(self, pts, layer_idx=(), _groups=None, _cut=None, _after_g2e=None):
        """Everything up to and including the encoder (partseg.py:527-545 / :407-425) -> (encoder output, group centres).
        _groups (an extension the trainer uses, not part of the reference signature): (neighborhood, center, event) computed by
        divide_patches on ANOTHER stream; the K / V producer -- which needs only the raw points -- is then issued first and this
        stream waits for the event only in front of Group2Emb, so FPS / kNN (latency-bound, a few workgroups) run beside it.
        _after_g2e (trainer extension, experiment VPF_IMG_GATE): called once Group2Emb's forward kernels are queued on this stream."""
        enc = self.encoder
        fuse_kv = (ops.adapter_kv_supported(self.input_adapter, pts) and enc.num_cross_attention_layers == 1 and pts.is_cuda
                   and ops.cfg.sa_fused and ops.cfg.enc_fused and self.training == enc.training)
        kv = None
        if not fuse_kv:
            pts_embs = self.input_adapter(pts)
        elif _groups is not None and pts.shape[0] * self.num_groups * pts.shape[2] > 0:
            cross = enc.cross_attn_1[0].module
            probe = pts.new_empty((pts.shape[0], self.num_groups, self.group2emb.second_conv[3].weight.shape[0]))
            if enc.fused_ok(probe, pts.new_empty((pts.shape[0], 1, 2 * probe.shape[2]))):
                params = list(self.input_adapter.parameters()) + list(cross.kv_norm.parameters()) + [cross.attention.k_proj.weight, cross.attention.v_proj.weight]
                kv = ops.AdapterKVFn.apply(pts, self.input_adapter, cross, *params)
        if callable(_groups):
            _groups = _groups()          # issued only now: the K / V producer's kernels above are the first nodes of this stream
        if _groups is not None:
            neighborhood, center, ev = _groups
            torch.cuda.current_stream().wait_event(ev)
        else:
            neighborhood, center = divide_patches(pts, self.num_groups, self.group_size)
        group_embs = self.group2emb(neighborhood)
        if _after_g2e is not None:
            _after_g2e()
        if ops.ca_front_supported(self.position_emb, group_embs, enc) and self.training == enc.training:
            # position MLP + (tokens + pos) + q_norm + q projection of the cross-attention layer in one kernel; the encoder picks up
            # what it needs from a one-shot stash on the layer (and computes it itself if anything about its inputs differs)
            pos_embs = ops.CaFrontFn.apply(center, self.position_emb, group_embs.detach(), enc, *self.position_emb.parameters())
        else:
            pos_embs = ops.PosMLPFn.apply(center, self.position_emb, *self.position_emb.parameters())
        if kv is not None:
            group_embs, pos_embs, kv = self._cut_here(_cut, group_embs, pos_embs, kv)
            return enc(group_embs, pos_embs, kv, layer_idx, kv_ready=True), center
        if fuse_kv:
            cross = enc.cross_attn_1[0].module
            probe = group_embs.new_empty((group_embs.shape[0], 1, 2 * group_embs.shape[2]))       # shape probe only
            if enc.fused_ok(group_embs, probe):
                # adapter -> kv LayerNorm -> K / V projection in one kernel: the per-point embedding is never read back
                params = list(self.input_adapter.parameters()) + list(cross.kv_norm.parameters()) + [cross.attention.k_proj.weight, cross.attention.v_proj.weight]
                kv = ops.AdapterKVFn.apply(pts, self.input_adapter, cross, *params)
                group_embs, pos_embs, kv = self._cut_here(_cut, group_embs, pos_embs, kv)
                return enc(group_embs, pos_embs, kv, layer_idx, kv_ready=True), center
            pts_embs = self.input_adapter(pts)
        group_embs, pos_embs, pts_embs = self._cut_here(_cut, group_embs, pos_embs, pts_embs)
        return enc(group_embs, pos_embs, pts_embs, layer_idx), center


class CrossFormer_pc_mp(_PointBackbone):
    """partseg.py:473-550: point-cloud branch of --mp pre-training.
    forward(pts [B,N,3]) -> (feats [B,D], backbone [B,2D])."""

    def __init__(self, input_adapter=None, num_latents=128, num_latent_channels=384, group_size=32,
                 num_cross_attention_layers=1, num_cross_attention_heads=6, num_self_attention_layers=6,
                 num_self_attention_heads=6, mlp_widen_factor=4, max_dpr=.0, atten_drop=0.1, mlp_drop=.5, modal_prior=True):
        super().__init__()
        self.num_groups = num_latents
        self.group_size = group_size
        self.group2emb = Group2Emb(num_latent_channels)
        self.position_emb = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, num_latent_channels))
        self.input_adapter = input_adapter
        self.encoder = _encoder(num_latent_channels, num_cross_attention_layers, num_cross_attention_heads,
                                num_self_attention_layers, num_self_attention_heads, mlp_widen_factor, max_dpr, atten_drop,
                                mlp_drop, modal_prior)
        self.latent_head = _latent_head(num_latent_channels)
        ops.assign_sites(self, "pc")          # dropout sites by module name: independent of what else the process built

    def forward(self, pts, _groups=None, _cut=None, _after_g2e=None):
        backbone_feats = self.backbone(pts, _groups, _cut, _after_g2e)
        feats = ops.HeadFn.apply(backbone_feats, self.latent_head, self.training, *self.latent_head.parameters())
        return feats, backbone_feats


class CrossFormer_pc_mp_ft(CrossFormer_pc_mp):
    """partseg.py:553-605: the ModelNet fine-tuning classifier = the pre-training backbone + a three-block BatchNorm - ReLU -
    Linear head.  forward(pts [B,N,3]) -> logits [B, num_obj_classes].  (latent_head stays a member, unused, as in the
    reference: a pre-trained pc_model_best.pth loads with strict=False either way.)"""

    def __init__(self, input_adapter=None, num_latents=128, num_latent_channels=384, group_size=32,
                 num_cross_attention_layers=1, num_cross_attention_heads=6, num_self_attention_layers=6,
                 num_self_attention_heads=6, mlp_widen_factor=4, max_dpr=0, atten_drop=0.1, mlp_drop=0.5, modal_prior=True,
                 num_obj_classes=40):
        super().__init__(input_adapter, num_latents, num_latent_channels, group_size, num_cross_attention_layers,
                         num_cross_attention_heads, num_self_attention_layers, num_self_attention_heads, mlp_widen_factor, max_dpr,
                         atten_drop, mlp_drop, modal_prior)
        D = num_latent_channels
        self.finetune_head = nn.Sequential(nn.BatchNorm1d(2 * D), nn.ReLU(), nn.Linear(2 * D, D),
                                           nn.BatchNorm1d(D), nn.ReLU(), nn.Linear(D, D // 2),
                                           nn.BatchNorm1d(D // 2), nn.ReLU(), nn.Linear(D // 2, num_obj_classes))

    # ft_cls.py:146,166,179-181 runs under autocast + GradScaler: the incoming gradient is already large there and the internal scale
    # (ops.internal_grad_scale: never below 1) is the identity; it matters for callers that run this model WITHOUT a scaler (fp32 loops)
    internal_grad_scale = True

    def forward(self, pts):
        with ops.internal_grad_scale(self.internal_grad_scale and self.training and torch.is_grad_enabled()) as sc:
            h = self.backbone(pts)
            head = self.finetune_head
            for i in (0, 3, 6):
                h = ops.BnReluLinearFn.apply(h, head[i], head[i + 2], self.training, *head[i].parameters(), *head[i + 2].parameters())
        return ops.ScaleGradFn.apply(h, sc) if sc.on else h


class CrossFormer_partseg(_PointBackbone):
    """partseg.py:345-470: ShapeNetPart part segmentation = the pre-training backbone with taps at the self-attention layers
    ``layer_idx`` (1-based) + LayerNorm + global max / mean + object-label branch + PointNetFeaturePropagation + a three-layer
    per-point head.  forward(pts [B,N,3], cls_label [B,16]) -> logits [B,N,num_part_classes].

    Same member names as the reference, so a hot-path ``pc_model_best.pth`` loads with strict=False exactly like
    ft_partseg.py:80-83 does it (the 12 ``latent_head.*`` keys are unexpected, the head's keys missing)."""

    def __init__(self, input_adapter=None, num_latents=128, num_latent_channels=384, group_size=32,
                 num_cross_attention_layers=1, num_cross_attention_heads=6, num_self_attention_layers=12,
                 num_self_attention_heads=6, mlp_widen_factor=4, max_dpr=0.1, atten_drop=.0, mlp_drop=.0, layer_idx=[],
                 num_part_classes=50):
        super().__init__()
        from .utils import PointNetFeaturePropagation
        D = num_latent_channels
        self.num_groups = num_latents
        self.group_size = group_size
        self.group2emb = Group2Emb(D)
        self.position_emb = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, D))
        self.input_adapter = input_adapter
        self.encoder = _encoder(D, num_cross_attention_layers, num_cross_attention_heads, num_self_attention_layers,
                                num_self_attention_heads, mlp_widen_factor, max_dpr, atten_drop, mlp_drop, False)
        self.layer_idx = layer_idx
        self.norm = nn.LayerNorm(D)
        self.label_conv = nn.Sequential(nn.Conv1d(16, 64, kernel_size=1, bias=False), nn.BatchNorm1d(64), nn.LeakyReLU(0.2))
        self.num_layer_idx = len(layer_idx)
        self.propagation = PointNetFeaturePropagation(in_channel=self.num_layer_idx * D + 3, mlp=[mlp_widen_factor * D, 1024])
        self.conv1 = nn.Conv1d(2 * self.num_layer_idx * D + 64 + 1024, 512, 1)
        self.bn1 = nn.BatchNorm1d(512)
        self.dp1 = nn.Dropout(0.5)
        self.dp1.site = ops.new_site()
        self.conv2 = nn.Conv1d(512, 256, 1)
        self.bn2 = nn.BatchNorm1d(256)
        self.conv3 = nn.Conv1d(256, num_part_classes, 1)
        self.relu = nn.ReLU()
        ops.assign_sites(self, "pcseg")

    # ft_partseg.py:145-176 has no GradScaler and averages its loss over B x N points (gradients of 1e-6 at the logits): the backward
    # pass of this model normalises the gradient that enters it (ops.internal_grad_scale; tools/diag_ft_scale.py)
    internal_grad_scale = True

    def forward(self, pts, cls_label):
        with ops.internal_grad_scale(self.internal_grad_scale and self.training and torch.is_grad_enabled()) as sc:
            y = self._forward(pts, cls_label)
        return ops.ScaleGradFn.apply(y, sc) if sc.on else y

    def _forward(self, pts, cls_label):
        from ... import ops_seg as S
        if self.num_layer_idx not in (3, 4):
            # partseg.py:430-435 only defines x for 3 or 4 taps (UnboundLocalError otherwise)
            raise ValueError("layer_idx must hold 3 or 4 layer numbers")
        feats, center = self._encode(pts, self.layer_idx)
        if len(feats) != self.num_layer_idx:
            raise ValueError(f"layer_idx {self.layer_idx} must be distinct values in 1..{len(self.encoder.sa_layers)}")
        nl = len(feats)
        xcat = S.LnTapsFn.apply(self.norm, nl, *feats, *self.norm.parameters())                     # [B,G,nl*D]
        pooled = ops.PoolFn.apply(xcat)                                                             # [B, 2*nl*D] = [x_max | x_avg]
        lf = S.LabelBranchFn.apply(cls_label, self.label_conv, self.training, *self.label_conv.parameters())
        prop = self.propagation
        f0 = S.FeaturePropFn.apply(pts, center, pts[:, :, :3], xcat, prop, self.training, *prop.parameters())
        gvec = torch.cat([pooled, lf], dim=1)                                                       # conv1's per-cloud input channels
        head = [self.conv1, self.bn1, self.conv2, self.bn2, self.conv3]
        return S.SegConvFn.apply(f0, gvec, self, self.training, *[p for m in head for p in m.parameters()])


class CrossFormer_img_mp(nn.Module):
    """partseg.py:608-680: image branch.  forward(imgs [B,H,W,3]) -> (feats [B,D], backbone [B,2D]).
    ``imgs`` may be the permuted NCHW view pretrain.py:179 hands over (strides are honoured)."""

    def __init__(self, img_height=144, img_width=144, patch_size=12, num_latent_channels=384,
                 num_cross_attention_layers=1, num_cross_attention_heads=6, num_self_attention_layers=6,
                 num_self_attention_heads=6, mlp_widen_factor=4, max_dpr=.0, atten_drop=0.1, mlp_drop=.5, modal_prior=True):
        super().__init__()
        num_patches = (img_height // patch_size) * (img_width // patch_size)
        self.patch_size = patch_size
        self.patch2emb = nn.Sequential(_Patchify(patch_size), nn.Linear(patch_size * patch_size * 3, num_latent_channels))
        self.position_emb = nn.Parameter(torch.randn(1, num_patches, num_latent_channels))
        self.encoder = _encoder(num_latent_channels, num_cross_attention_layers, num_cross_attention_heads,
                                num_self_attention_layers, num_self_attention_heads, mlp_widen_factor, max_dpr, atten_drop,
                                mlp_drop, modal_prior)
        self.latent_head = _latent_head(num_latent_channels)
        ops.assign_sites(self, "img")

    def forward(self, imgs):
        lin = self.patch2emb[1]
        patch_embs = ops.PatchEmbedFn.apply(imgs, lin, self.patch_size, *lin.parameters())
        if patch_embs.shape[1] != self.position_emb.shape[1]:
            raise ValueError(f"image gives {patch_embs.shape[1]} patches, position_emb has {self.position_emb.shape[1]}")
        x = self.encoder(patch_embs, self.position_emb, patch_embs)
        backbone_feats = ops.PoolFn.apply(x)
        feats = ops.HeadFn.apply(backbone_feats, self.latent_head, self.training, *self.latent_head.parameters())
        return feats, backbone_feats
