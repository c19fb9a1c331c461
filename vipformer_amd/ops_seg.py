"""Host-side glue of the part-segmentation head (BASELINE config 5): CrossFormer_partseg.forward partseg.py:407-470 and
PointNetFeaturePropagation.forward utils.py:205-242 as autograd Functions over the C ABI -- the same conventions as ops.py (fp32
master weights, h16 MFMA operands, fp32 accumulation and statistics; weight gradients through ops._GradSink).

    taps (encoder layers layer_idx, fp32 [B,G,D] each)
      -> LnTapsFn      : LayerNorm (shared parameters) of every tap, concatenated        -> xcat fp32 [B,G,F]   F = len(layer_idx) * D
      -> ops.PoolFn    : cat[max over groups, mean over groups]                          -> [B,2F]
    cls_label [B,16] -> LabelBranchFn : conv1d(16,64,no bias) + BatchNorm + LeakyReLU(0.2) -> [B,64]
    FeaturePropFn(pts, centers, points1 = pts, xcat): 3-NN inverse-distance interpolation + (conv1d + BatchNorm + ReLU) x len(mlp)
                                                                                         -> f_level_0 h16 [B,N,1024]
    SegConvFn(f_level_0, [max | mean | label]) : conv1 on cat(f_level_0, global.repeat(N)) -- the global part enters as a
        per-cloud bias (global . W[:,1024:]^T + b), it is never repeated N times -- BatchNorm ReLU Dropout(0.5) conv2 BatchNorm ReLU conv3
                                                                                         -> logits fp32 [B,N,num_part_classes]
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import ops
from .ops import (H16, EPI_ATOMIC, EPI_GROUPBIAS, F32, _bn_act, _bn_bwd, _bn_stat, _sinked, colsum, gemm, grad_buf, linear_dgrad,
                  linear_fwd, linear_wgrad, shadow, to_h16)

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2        # the `relu` argument of the BatchNorm entry points (2 = LeakyReLU(0.2))


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def _pad_h16(src, rows, K, rows_out, Kp):
    """h16 [rows_out, Kp] zero-padded copy of src [rows, K] (vpf_pad_h16): GEMM operands need multiples of 8."""
    out = torch.empty(rows_out, Kp, dtype=H16, device=src.device)
    L.call("vpf_pad_h16", src, int(src.dtype == H16), rows, K, K, rows_out, Kp, out)
    return out


class LnTapsFn(torch.autograd.Function):
    """partseg.py:427-435: x = cat([self.norm(t) for t in taps], channels) -> fp32 [B,G,nl*D] (row-major: channel block i = tap i)."""

    @staticmethod
    def forward(ctx, norm, nl, *rest):
        taps, params = rest[:nl], rest[nl:]
        ctx.nparams, ctx.params = len(params), params
        if not 1 <= nl <= 4:
            raise L.VpfError("layer_idx must name 1..4 encoder layers (the reference accepts 3 or 4, partseg.py:430-435)")
        B, G, D = taps[0].shape
        rows = B * G
        xs = [t.contiguous().float() for t in taps]
        dev = xs[0].device
        xcat = torch.empty(B, G, nl * D, dtype=F32, device=dev)
        mean = torch.empty(nl * rows, dtype=F32, device=dev)
        rstd = torch.empty(nl * rows, dtype=F32, device=dev)
        p4 = xs + [None] * (4 - nl)
        L.call("vpf_ln_taps_fwd", p4[0], p4[1], p4[2], p4[3], nl, rows, D, norm.weight.data, norm.bias.data, float(norm.eps), xcat, mean, rstd)
        ctx.norm, ctx.nl, ctx.dims = norm, nl, (B, G, D)
        ctx.save_for_backward(mean, rstd, *xs)
        return xcat

    @staticmethod
    @_sinked
    def backward(ctx, dxcat):
        mean, rstd, *xs = ctx.saved_tensors
        norm, nl = ctx.norm, ctx.nl
        B, G, D = ctx.dims
        rows = B * G
        dxcat = dxcat.contiguous().float()
        ds = [torch.empty(B, G, D, dtype=F32, device=dxcat.device) for _ in range(nl)]
        p4, d4 = xs + [None] * (4 - nl), ds + [None] * (4 - nl)
        L.call("vpf_ln_taps_bwd", dxcat, p4[0], p4[1], p4[2], p4[3], nl, rows, D, mean, rstd, norm.weight.data, d4[0], d4[1], d4[2], d4[3],
               grad_buf(norm.weight), grad_buf(norm.bias))
        return (None, None) + tuple(ds) + (None,) * ctx.nparams


class LabelBranchFn(torch.autograd.Function):
    """partseg.py:391-393,447-450: label_conv = Conv1d(16,64,1,bias=False) + BatchNorm1d(64) + LeakyReLU(0.2) on the one-hot object
    class [B,16,1] (BatchNorm over the B samples) -> fp32 [B,64]."""

    @staticmethod
    def forward(ctx, cls_label, seq, training, *params):
        ctx.nparams, ctx.params = len(params), params
        conv, bn = seq[0], seq[1]
        Cin, Cout = conv.weight.shape[1], conv.weight.shape[0]
        B = cls_label.shape[0]
        lab = cls_label.reshape(B, Cin)
        L.need_cuda(lab)
        lab16 = to_h16(lab)
        z = linear_fwd(lab16, shadow([conv.weight]), Cout, Cin, None, out_f32=True)
        small = training and B <= 4096 and Cout % 64 == 0
        y = torch.empty(B, Cout, dtype=H16, device=z.device)
        if small:
            st = torch.empty(2 * Cout, dtype=F32, device=z.device)
            L.call("vpf_bn_small_fwd", z, B, Cout, bn.weight.data, bn.bias.data, float(bn.eps), float(bn.momentum), bn.running_mean,
                   bn.running_var, bn.num_batches_tracked, st, y, ACT_LEAKY)
        else:
            st = _bn_stat(z, Cout, bn, training)
            y = _bn_act(z, Cout, st, bn, ACT_LEAKY, True)
        ctx.mods, ctx.training, ctx.small = (conv, bn), training, small
        ctx.save_for_backward(lab16, z, st)
        return ops.to_f32(y)

    @staticmethod
    @_sinked
    def backward(ctx, dy):
        lab16, z, st = ctx.saved_tensors
        conv, bn = ctx.mods
        B, Cout = z.shape
        Cin = lab16.shape[1]
        dy = dy.contiguous().float()
        if ctx.small:
            dz = torch.empty(B, Cout, dtype=H16, device=z.device)
            L.call("vpf_bn_small_bwd", dy, z, st, bn.weight.data, bn.bias.data, B, Cout, ACT_LEAKY, dz, 1, grad_buf(bn.weight), grad_buf(bn.bias))
        else:
            dz = _bn_bwd(dy, z, Cout, st, bn, ACT_LEAKY, ctx.training, True)
        linear_wgrad(dz, lab16, Cout, Cin, grad_buf(conv.weight))
        return (None, None, None) + (None,) * ctx.nparams


def _conv_bn_relu_fwd(x16, K, conv, bn, training, w16=None):
    """relu(bn(conv1d_k1(x))) on rows: x16 h16 [M,K] -> (h f32 [M,Cout] pre-norm, stat, a h16 [M,Cout]).  The pre-normalisation
    activation stays fp32: training-mode BatchNorm divides by the spread of a channel over the rows, which for smoothly interpolated
    features is a small fraction of its magnitude -- a h16 rounding of h (relative to |h|) would be amplified by |h| / std(h)."""
    Cout = conv.weight.shape[0]
    h = linear_fwd(x16, w16 if w16 is not None else shadow([conv.weight]), Cout, K, conv.bias.data if conv.bias is not None else None,
                   out_f32=True)
    st = _bn_stat(h, Cout, bn, training)
    return h, st, _bn_act(h, Cout, st, bn, ACT_RELU, True)


class FeaturePropFn(torch.autograd.Function):
    """PointNetFeaturePropagation.forward (utils.py:205-242) in row-major layouts: xyz1 [B,N,3+] (targets), xyz2 [B,S,3+] (sources),
    points1 [B,N,C1] or None (concatenated IN FRONT of the interpolated features, :232-236), feat fp32 [B,S,F] (= points2) ->
    h16 [B,N,mlp[-1]].  The full sort of utils.py:224 is a running 3-minimum (vpf_three_nn_f32), the gather + weighting + concat
    land directly in the first convolution's operand (vpf_interp_rows_fwd)."""

    @staticmethod
    def forward(ctx, xyz1, xyz2, points1, feat, mod, training, *params):
        ctx.nparams, ctx.params = len(params), params
        L.need_cuda(xyz1, xyz2, feat)
        B, N, _ = xyz1.shape
        S, Fd = xyz2.shape[1], feat.shape[2]
        if S == 2:
            # utils.py:224-230: dists[:, :, :3] of two centres has two columns and weight.view(B, N, 3, 1) raises
            raise RuntimeError("PointNetFeaturePropagation: three neighbours cannot be taken from S = 2 centres (the reference raises here too)")
        dev = feat.device
        xyz1c, xyz2c = xyz1.detach().contiguous().float(), xyz2.detach().contiguous().float()
        featc = feat.contiguous().float()
        M = B * N
        idx = torch.empty(M * 3, dtype=torch.int32, device=dev)
        w = torch.empty(M * 3, dtype=F32, device=dev)
        L.call("vpf_three_nn_f32", xyz1c, B, N, xyz1c.shape[2], xyz2c, xyz2c.shape[2], S, idx, w)
        if points1 is not None:
            p1 = points1.detach().contiguous().float()
            C1 = p1.shape[2]
        else:
            p1, C1 = xyz1c, 0
        Kin = C1 + Fd
        Kp = _pad8(Kin)
        A0 = torch.empty(M, Kp, dtype=H16, device=dev)
        L.call("vpf_interp_rows_fwd", featc, p1, B, N, C1, S, Fd, idx, w, Kp, A0)
        convs, bns = list(mod.mlp_convs), list(mod.mlp_bns)
        if convs[0].weight.shape[1] != Kin:
            raise L.VpfError(f"PointNetFeaturePropagation: in_channel {convs[0].weight.shape[1]} != {C1} + {Fd}")
        w0p = _pad_h16(shadow([convs[0].weight]), convs[0].weight.shape[0], Kin, convs[0].weight.shape[0], Kp)
        saved, x16, K = [], A0, Kp
        for i, (conv, bn) in enumerate(zip(convs, bns)):
            h, st, a = _conv_bn_relu_fwd(x16, K, conv, bn, training, w0p if i == 0 else None)
            saved += [x16, h, st]
            x16, K = a, conv.weight.shape[0]
        ctx.mod, ctx.training, ctx.dims = mod, training, (B, N, S, Fd, C1, Kp)
        ctx.w0p = w0p
        ctx.save_for_backward(idx, w, *saved)
        return x16.view(B, N, K)

    @staticmethod
    @_sinked
    def backward(ctx, dout):
        idx, w, *saved = ctx.saved_tensors
        mod, training = ctx.mod, ctx.training
        B, N, S, Fd, C1, Kp = ctx.dims
        convs, bns = list(mod.mlp_convs), list(mod.mlp_bns)
        M = B * N
        d = to_h16(dout).view(M, -1)
        for i in range(len(convs) - 1, -1, -1):
            conv, bn = convs[i], bns[i]
            x16, h, st = saved[3 * i:3 * i + 3]
            Cout = conv.weight.shape[0]
            dh = _bn_bwd(d, h, Cout, st, bn, ACT_RELU, training, True)
            if i > 0:
                K = conv.weight.shape[1]
                linear_wgrad(dh, x16, Cout, K, grad_buf(conv.weight), grad_buf(conv.bias) if conv.bias is not None else None)
                d = linear_dgrad(dh, shadow([conv.weight]), Cout, K)
            else:
                Kin = C1 + Fd
                dWp = torch.zeros(Cout, Kp, dtype=F32, device=dh.device)           # gradient of the zero-padded operand copy
                linear_wgrad(dh, x16, Cout, Kp, dWp, grad_buf(conv.bias) if conv.bias is not None else None)
                grad_buf(conv.weight).view(Cout, Kin).add_(dWp[:, :Kin])
                d = linear_dgrad(dh, ctx.w0p, Cout, Kp)
        dfeat = None
        if ctx.needs_input_grad[3]:
            dfeat = torch.zeros(B, S, Fd, dtype=F32, device=d.device)
            L.call("vpf_interp_rows_bwd", d, B, N, C1, S, Fd, idx, w, Kp, dfeat)
        return (None, None, None, dfeat, None, None) + (None,) * ctx.nparams


class SegConvFn(torch.autograd.Function):
    """partseg.py:452-468: x = cat(f_level_0, x_global_feature.repeat(N)); relu(bn1(conv1(x))); dp1; relu(bn2(conv2)); conv3; permute.
    f0 h16 [B,N,C0]; gvec fp32 [B,Cg] = [x_max | x_avg | label feature] (the order of conv1's input channels after f_level_0)."""

    @staticmethod
    def forward(ctx, f0, gvec, mod, training, *params):
        ctx.nparams, ctx.params = len(params), params
        B, N, C0 = f0.shape
        Cg = gvec.shape[1]
        dev = f0.device
        M = B * N
        conv1, bn1, conv2, bn2, conv3 = mod.conv1, mod.bn1, mod.conv2, mod.bn2, mod.conv3
        Kt, C1o = conv1.weight.shape[1], conv1.weight.shape[0]
        if Kt != C0 + Cg or C0 % 8 or Cg % 8:
            raise L.VpfError(f"conv1 expects {Kt} input channels, got {C0} + {Cg} (both multiples of 8)")
        f016 = to_h16(f0).view(M, C0)
        g16 = to_h16(gvec)
        w1 = shadow([conv1.weight])                                               # [C1o, Kt]: columns [0,C0) per point, [C0,Kt) per cloud
        gb = torch.empty(B, C1o, dtype=F32, device=dev)
        gemm(g16, 0, Cg, w1[C0:], 0, Kt, B, C1o, Cg, gb, C1o, c_f32=True, bias=conv1.bias.data)      # the per-cloud part, once per cloud
        h3 = torch.empty(M, C1o, dtype=F32, device=dev)                            # pre-BatchNorm activations stay fp32 (see _conv_bn_relu_fwd)
        gemm(f016, 0, C0, w1, 0, Kt, M, C1o, C0, h3, C1o, c_f32=True, mode=EPI_GROUPBIAS, gbias=gb, group=N)
        st1 = _bn_stat(h3, C1o, bn1, training)
        a3 = _bn_act(h3, C1o, st1, bn1, ACT_RELU, True)
        p = float(mod.dp1.p) if training else 0.0
        ctx.rng_st = ops.rng.acquire(dev, p > 0.0)
        if p > 0.0:
            tmp = torch.empty(M, C1o, dtype=F32, device=dev)
            L.call("vpf_dropout_add_fwd", a3, None, tmp, tmp.numel(), ctx.rng_st, mod.dp1.site, p)
            a3d = to_h16(tmp)
        else:
            a3d = a3
        C2o = conv2.weight.shape[0]
        h4 = linear_fwd(a3d, shadow([conv2.weight]), C2o, C1o, conv2.bias.data, out_f32=True)
        st2 = _bn_stat(h4, C2o, bn2, training)
        a4 = _bn_act(h4, C2o, st2, bn2, ACT_RELU, True)
        NC = conv3.weight.shape[0]
        NCp = _pad8(NC)
        w3p = _pad_h16(shadow([conv3.weight]), NC, C2o, NCp, C2o)
        b3p = torch.zeros(NCp, dtype=F32, device=dev)
        b3p[:NC] = conv3.bias.data
        logits = linear_fwd(a4, w3p, NCp, C2o, b3p, out_f32=True)
        ctx.mod, ctx.training, ctx.dims, ctx.p, ctx.w3p = mod, training, (B, N, C0, Cg, NC, NCp), p, w3p
        ctx.f0_dtype = f0.dtype
        ctx.save_for_backward(f016, g16, h3, st1, a3d, h4, st2, a4)
        return logits.view(B, N, NCp)[:, :, :NC]

    @staticmethod
    @_sinked
    def backward(ctx, dlogits):
        f016, g16, h3, st1, a3d, h4, st2, a4 = ctx.saved_tensors
        mod, training, p = ctx.mod, ctx.training, ctx.p
        B, N, C0, Cg, NC, NCp = ctx.dims
        conv1, bn1, conv2, bn2, conv3 = mod.conv1, mod.bn1, mod.conv2, mod.bn2, mod.conv3
        Kt, C1o, C2o = conv1.weight.shape[1], conv1.weight.shape[0], conv2.weight.shape[0]
        M = B * N
        dev = dlogits.device
        dl16 = _pad_h16(dlogits.contiguous().float().view(M, NC), M, NC, M, NCp)
        dW3 = torch.zeros(NCp, C2o, dtype=F32, device=dev)
        db3 = torch.zeros(NCp, dtype=F32, device=dev)
        linear_wgrad(dl16, a4, NCp, C2o, dW3, db3)
        grad_buf(conv3.weight).view(NC, C2o).add_(dW3[:NC])
        grad_buf(conv3.bias).add_(db3[:NC])
        da4 = linear_dgrad(dl16, ctx.w3p, NCp, C2o)
        dh4 = _bn_bwd(da4, h4, C2o, st2, bn2, ACT_RELU, training, True)
        linear_wgrad(dh4, a3d, C2o, C1o, grad_buf(conv2.weight), grad_buf(conv2.bias))
        if p > 0.0:
            da3f = linear_dgrad(dh4, shadow([conv2.weight]), C2o, C1o, out_f32=True)
            da3 = torch.empty(M, C1o, dtype=H16, device=dev)
            L.call("vpf_dropout_bwd", da3f, da3, da3f.numel(), ctx.rng_st, mod.dp1.site, p)
        else:
            da3 = linear_dgrad(dh4, shadow([conv2.weight]), C2o, C1o)
        dh3 = _bn_bwd(da3, h3, C1o, st1, bn1, ACT_RELU, training, True)
        w1 = shadow([conv1.weight])
        gW1 = grad_buf(conv1.weight).view(-1)                                      # [C1o, Kt] row-major
        gemm(dh3, 1, C1o, f016, 1, C0, C1o, C0, M, gW1, Kt, c_f32=True, mode=EPI_ATOMIC)              # dW1[:, :C0]
        dgb = torch.zeros(B, C1o, dtype=F32, device=dev)                                              # d(per-cloud bias) = sum over the cloud's points
        for b in range(B):
            colsum(dh3[b * N:(b + 1) * N], C1o, dgb[b])
        colsum(dgb, C1o, grad_buf(conv1.bias))
        dgb16 = to_h16(dgb)
        gemm(dgb16, 1, C1o, g16, 1, Cg, C1o, Cg, B, gW1[C0:], Kt, c_f32=True, mode=EPI_ATOMIC)       # dW1[:, C0:]
        dg = torch.empty(B, Cg, dtype=F32, device=dev)
        gemm(dgb16, 0, C1o, w1[C0:], 1, Kt, B, Cg, C1o, dg, Cg, c_f32=True)
        df0 = torch.empty(M, C0, dtype=H16, device=dev)
        gemm(dh3, 0, C1o, w1, 1, Kt, M, C0, C1o, df0, C0, c_f32=False)
        df0 = df0.view(B, N, C0)
        return (df0 if ctx.f0_dtype == H16 else ops.to_f32(df0), dg, None, None) + (None,) * ctx.nparams


class CrossEntropySmoothFn(torch.autograd.Function):
    """torch.nn.CrossEntropyLoss(label_smoothing=eps) of ft_partseg.py:128,155 (mean over rows), forward + gradient in one kernel."""

    @staticmethod
    def forward(ctx, logits, target, eps):
        C = logits.shape[-1]
        z = logits.reshape(-1, C)
        if z.stride(1) != 1 or z.dtype != F32:
            z = z.contiguous().float()
        rows = z.shape[0]
        t = target.reshape(-1).contiguous().long()
        dev = z.device
        loss = torch.empty(1, dtype=F32, device=dev)
        dz = torch.empty(rows, C, dtype=F32, device=dev)
        L.call("vpf_ce_smooth", z, z.stride(0), t, rows, C, float(eps), torch.empty(1024, dtype=F32, device=dev), loss, dz, C)
        ctx.shape = logits.shape
        ctx.save_for_backward(dz)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        (dz,) = ctx.saved_tensors
        return (dz * dloss).view(ctx.shape), None, None


def cross_entropy_smooth(logits, target, label_smoothing=0.2):
    return CrossEntropySmoothFn.apply(logits, target, label_smoothing)


from .ops import _scale_aware_forwards as _saf  # noqa: E402
_saf(globals())
