"""GPU parity of the individual HIP kernels (through the C ABI) against plain torch fp32
references of the same op on the same h16-rounded operands.  Tolerances are stated per test."""
import math

import numpy as np
import pytest
import torch

from tests import helpers as Hh
from tests.helpers import H16

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def rnd(seed, *shape, scale=1.0):
    return (Hh.synth_like(seed, shape) * scale).cuda()


def bf(x):
    return x.to(H16)


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("a_tr,b_tr", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 64, 32), (64, 64, 64), (200, 72, 40), (1000, 256, 256), (96, 136, 520), (8, 8, 8)])
def test_gemm_layouts_exact_integers(a_tr, b_tr, M, N, K):
    """Exact small-integer data, ASYMMETRIC operands: catches any row/col or k-order mix-up of the
    MFMA fragment maps and of the transposing LDS reads (results must be bit-exact)."""
    from vipformer_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    B = torch.randint(-3, 4, (N, K), generator=g).float()
    A[0, 0], A[M - 1, K - 1], B[0, K - 1], B[N - 1, 0] = 5, -7, 6, -5
    ref = A @ B.t()
    Ad = bf(A.t().contiguous() if a_tr else A).cuda()
    Bd = bf(B.t().contiguous() if b_tr else B).cuda()
    C = torch.empty(M, N, dtype=torch.float32, device="cuda")
    ops.gemm(Ad, a_tr, M if a_tr else K, Bd, b_tr, N if b_tr else K, M, N, K, C, N, c_f32=True)
    assert torch.equal(C.cpu(), ref), f"max abs diff {(C.cpu() - ref).abs().max()}"


def test_gemm_epilogues_and_splitk():
    from vipformer_amd import ops
    M, N, K = 520, 264, 136
    A, W, bias = rnd(1, M, K), rnd(2, N, K, scale=0.1), rnd(3, N)
    A16, W16 = bf(A), bf(W)
    ref = A16.float() @ W16.float().t() + bias
    y = ops.linear_fwd(A16, W16, N, K, bias)
    assert rel(y.float(), ref) < 4e-3          # h16 output rounding
    y32 = ops.linear_fwd(A16, W16, N, K, bias, out_f32=True)
    assert rel(y32, ref) < 1e-5                # fp32 accumulate, only summation order differs
    # GELU epilogue (+ pre-activation), GELU' dgrad epilogue
    u = torch.empty(M, N, dtype=H16, device="cuda")
    h = ops.linear_fwd(A16, W16, N, K, bias, mode=ops.EPI_GELU, C2=u, ldc2=N)
    assert rel(u.float(), ref) < 4e-3
    assert rel(h.float(), torch.nn.functional.gelu(u.float())) < 4e-3
    dY = bf(rnd(4, M, N))
    dx = ops.linear_dgrad(dY, W16, N, K, out_f32=True)
    assert rel(dx, dY.float() @ W16.float()) < 1e-5
    Wk = bf(rnd(5, K, N, scale=0.1))           # dgrad through a [K_out=N... shape check with GELU'
    du = ops.linear_dgrad(bf(rnd(6, M, K)), Wk, K, N, mode=ops.EPI_GELU_BWD, aux=u, ldaux=N)
    uu = u.float().requires_grad_()
    torch.nn.functional.gelu(uu).backward(bf(rnd(6, M, K)).float() @ Wk.float())
    assert rel(du.float(), uu.grad) < 5e-3
    # wgrad: split-K atomics accumulate INTO the buffer
    dW = torch.ones(N, K, dtype=torch.float32, device="cuda")
    ops.linear_wgrad(dY, A16, N, K, dW)
    assert rel(dW, 1.0 + dY.float().t() @ A16.float()) < 1e-5
    # dropout + residual epilogue against the exported mask
    res = rnd(7, M, N)
    site, p = 12345, 0.3
    out = ops.linear_fwd(A16, W16, N, K, bias, out_f32=True, mode=ops.EPI_DROP_RES, res=res, ldres=N, site=site, p=p)
    keep = ops.dropout_keep_mask(site, p, (M, N), "cuda").float()
    assert abs(keep.mean().item() - (1 - p)) < 0.01
    assert rel(out, res + ref * keep / (1 - p)) < 1e-5
    # group bias + batched
    gb = rnd(8, M // 8, N)
    yg = torch.empty(M, N, dtype=torch.float32, device="cuda")
    ops.gemm(A16, 0, K, W16, 0, K, M, N, K, yg, N, c_f32=True, mode=ops.EPI_GROUPBIAS, gbias=gb, group=8)
    assert rel(yg, A16.float() @ W16.float().t() + gb.repeat_interleave(8, 0)) < 1e-5
    Ab = bf(rnd(9, 3, 64, 40)); Bb = bf(rnd(10, 3, 48, 40))
    Cb = torch.empty(3, 64, 48, dtype=torch.float32, device="cuda")
    ops.gemm(Ab, 0, 40, Bb, 0, 40, 64, 48, 40, Cb, 48, c_f32=True, batch=3, sAb=64 * 40, sBb=48 * 40, sCb=64 * 48)
    assert rel(Cb, torch.bmm(Ab.float(), Bb.float().transpose(1, 2))) < 1e-5


def test_gemm_bad_alignment_is_loud():
    from vipformer_amd import _lib, ops
    A, B = bf(rnd(1, 16, 12)), bf(rnd(2, 16, 12))
    with pytest.raises(_lib.VpfError):
        ops.gemm(A, 0, 12, B, 0, 12, 16, 16, 12, torch.empty(16, 16, device="cuda"), 16, c_f32=True)


# ------------------------------------------------------------------------------------------ LayerNorm / dropout
@pytest.mark.parametrize("rows,D", [(1000, 256), (37, 64), (513, 384), (64, 128)])
def test_layernorm_fwd_bwd(rows, D):
    from vipformer_amd import ops
    x, pos = rnd(1, rows, D, scale=2.0), rnd(2, rows, D)
    gamma = torch.nn.Parameter(rnd(3, D) * 0.2 + 1.0); beta = torch.nn.Parameter(rnd(4, D) * 0.1)
    y, mean, rstd, xsum = ops.layernorm_fwd(x, gamma.data, beta.data, pos=pos, want_sum=True)
    xr = (x + pos).requires_grad_()
    g2, b2 = gamma.detach().clone().requires_grad_(), beta.detach().clone().requires_grad_()
    yr = torch.nn.functional.layer_norm(xr, (D,), g2, b2, 1e-5)
    assert torch.equal(xsum, x + pos) and rel(y.float(), yr) < 4e-3
    dy = bf(rnd(5, rows, D)); dres = rnd(6, rows, D)
    dx = ops.layernorm_bwd(dy, xsum, mean, rstd, gamma, beta, dres)
    yr.backward(dy.float())
    assert rel(dx, xr.grad + dres) < 1e-4 and rel(gamma.grad, g2.grad) < 1e-4 and rel(beta.grad, b2.grad) < 1e-4
    # broadcast pos ([T,D] over the batch) and h16 input
    T = rows // 4 if rows % 4 == 0 else rows
    y2, _, _, xs2 = ops.layernorm_fwd(x, gamma.data, beta.data, pos=pos[:T].contiguous(), want_sum=True)
    assert torch.equal(xs2, x + pos[:T].repeat(rows // T, 1))
    y3, m3, r3, _ = ops.layernorm_fwd(bf(x), gamma.data, beta.data)
    assert rel(y3.float(), torch.nn.functional.layer_norm(bf(x).float(), (D,), gamma.data, beta.data, 1e-5)) < 4e-3


def test_dropout_add_and_mask_statistics():
    from vipformer_amd import ops
    n = 1 << 20
    y, res = bf(rnd(1, n)), rnd(2, n)
    for p in (0.1, 0.5):
        site = ops.new_site()
        with ops.rng.pinned():               # draw from the process state itself, so the exported mask is the one used
            out = ops.DropoutAddFn.apply(y, res, p, site)
        keep = ops.dropout_keep_mask(site, p, (n,), "cuda").float()
        assert abs(keep.mean().item() - (1 - p)) < 3e-3                    # ~3 sigma at n = 1M is 1.5e-3
        assert torch.allclose(out, res + y.float() * keep / (1 - p), rtol=1e-6, atol=1e-6)
        # different sites / steps decorrelate
        k2 = ops.dropout_keep_mask(site + 1, p, (n,), "cuda").float()
        assert abs(((keep - keep.mean()) * (k2 - k2.mean())).mean().item()) < 2e-3
    ops.rng.advance("cuda")
    k3 = ops.dropout_keep_mask(site, 0.5, (n,), "cuda").float()
    assert (k3 != keep).float().mean().item() > 0.4


# ------------------------------------------------------------------------------------------ attention
def attn_ref(q, k, v, scale, keep, p):
    """q [B,H,Lq,64] etc. fp32; keep [B,H,Lq,Lkv] or None."""
    s = torch.einsum("bhid,bhjd->bhij", q, k) * scale
    a = s.softmax(-1)
    if keep is not None:
        a = a * keep / (1 - p)
    return torch.einsum("bhij,bhjd->bhid", a, v), torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,H,Lq,Lkv,p", [(2, 2, 96, 1024, 0.0), (2, 4, 196, 196, 0.0), (1, 1, 16, 8, 0.0), (2, 1, 33, 70, 0.0),
                                          (2, 2, 96, 96, 0.1), (2, 4, 128, 128, 0.1), (3, 2, 120, 120, 0.0), (1, 2, 128, 1024, 0.1), (2, 4, 196, 196, 0.1), (3, 1, 50, 200, 0.5), (1, 2, 300, 520, 0.1), (2, 1, 160, 96, 0.0),
                                          (2, 4, 144, 144, 0.1), (1, 2, 150, 150, 0.0), (2, 4, 96, 2048, 0.1)])      # 144: the reference scripts' 144 x 144 / patch 12 image (resident <5>); 96 x 2048: their clouds
def test_attention_fwd_bwd(B, H, Lq, Lkv, p):
    from vipformer_amd import _lib as L
    from vipformer_amd import ops
    D = 64 * H
    q, k, v = rnd(1, B, Lq, D), rnd(2, B, Lkv, D), rnd(3, B, Lkv, D)
    do = rnd(4, B, Lq, D)
    q16, k16, v16, do16 = bf(q), bf(k), bf(v), bf(do)
    site = ops.new_site()
    scale = 64 ** -0.5
    o = torch.empty(B * Lq, D, dtype=H16, device="cuda")
    lse = torch.empty(B * H * Lq, dtype=torch.float32, device="cuda")
    st = ops.rng.state("cuda")
    L.call("vpf_attention_fwd", q16, D, k16, D, v16, D, B, H, Lq, Lkv, 64, scale, p, st, site, o, D, lse)
    keep = ops.dropout_keep_mask(site, p, (B, H, Lq, Lkv), "cuda").float() if p > 0 else None
    split = lambda t, Lx: t.float().view(B, Lx, H, 64).permute(0, 2, 1, 3).contiguous().requires_grad_()
    qr, kr, vr = split(q16, Lq), split(k16, Lkv), split(v16, Lkv)
    oref, lref = attn_ref(qr, kr, vr, scale, keep, p)
    og = o.float().view(B, Lq, H, 64).permute(0, 2, 1, 3)
    assert rel(og, oref) < 8e-3, f"fwd rel {rel(og, oref)}"            # h16 P and h16 output
    assert torch.allclose(lse.view(B, H, Lq), lref, rtol=1e-4, atol=1e-4)
    dq = torch.empty(B * Lq, D, dtype=H16, device="cuda")
    dk = torch.empty(B * Lkv, D, dtype=H16, device="cuda")
    dv = torch.empty(B * Lkv, D, dtype=H16, device="cuda")
    L.call("vpf_attention_bwd", q16, D, k16, D, v16, D, o, D, do16, D, lse, B, H, Lq, Lkv, 64, scale, p, st, site,
           dq, D, dk, D, dv, D, torch.empty(B * H * Lq, dtype=torch.float32, device="cuda"))
    oref.backward(do16.float().view(B, Lq, H, 64).permute(0, 2, 1, 3))
    unsplit = lambda t, Lx: t.permute(0, 2, 1, 3).reshape(B * Lx, D)
    for name, got, ref in (("dq", dq, unsplit(qr.grad, Lq)), ("dk", dk, unsplit(kr.grad, Lkv)), ("dv", dv, unsplit(vr.grad, Lkv))):
        r = rel(got.float(), ref)
        assert r < 1.5e-2, f"{name} rel {r}"                                 # h16 P/dS operands + h16 outputs


@pytest.mark.parametrize("B,L,D", [(128, 96, 256), (3, 50, 384), (2, 7, 6), (2, 33, 1024), (5, 196, 256)])
def test_pool_fwd_bwd_vs_torch(B, L, D):
    """vpf_pool_fwd / vpf_pool_bwd (cat[max over tokens, mean over tokens], partseg.py:548): values, first-maximum argmax and the
    backward (mean's share to every token + the maximum's to its winner) against the same formula in torch (one division and one addition
    per element)."""
    from vipformer_amd import _lib as L_
    x = rnd(1, B, L, D).contiguous()
    out = torch.empty(B, 2 * D, device="cuda")
    arg = torch.empty(B, D, dtype=torch.int32, device="cuda")
    L_.call("vpf_pool_fwd", x, B, L, D, out, arg)
    xr = x.clone().requires_grad_()
    ref = torch.cat([xr.max(dim=1)[0], xr.mean(dim=1)], dim=1)
    assert torch.equal(out[:, :D], ref[:, :D].detach()) and torch.allclose(out[:, D:], ref[:, D:].detach(), rtol=1e-5, atol=1e-6)
    assert torch.equal(arg.long(), x.argmax(dim=1))
    dout = rnd(2, B, 2 * D).contiguous()
    dx = torch.full((B, L, D), float("nan"), device="cuda")
    L_.call("vpf_pool_bwd", dout, arg, B, L, D, dx)
    want = (dout[:, None, D:] / float(L)).expand(B, L, D).clone()
    want.scatter_add_(1, arg.long()[:, None, :], dout[:, None, :D])
    assert torch.allclose(dx, want, rtol=2e-6, atol=1e-9), float((dx - want).abs().max())     # (the kernel's division is not IEEE-rounded)


@pytest.mark.parametrize("B,H,Lq,Lkv,p", [(2, 2, 96, 1024, 0.1), (1, 2, 128, 1024, 0.1), (2, 1, 80, 1000, 0.1), (1, 1, 70, 518, 0.1),
                                          (1, 2, 96, 512, 0.0), (3, 1, 33, 640, 0.5)])
def test_attention_bwd_one_kernel_equals_two_bitwise(B, H, Lq, Lkv, p):
    """attn_bwd_ca_kernel (few queries, many keys: dQ, dK, dV in one kernel, the queries resident, dQ accumulated over key tiles by the
    wave that owns the query block) against attn_bwd_dq_kernel + attn_bwd_dkv_resq_kernel (VPF_ATTN_CA_MERGED = 0): same products in
    the same order at the same rounding points -- bit-identical gradients, ragged query and key counts and the unaligned-dropout
    fallback (Lkv % 4 != 0) included."""
    from vipformer_amd import _lib as L
    from vipformer_amd import ops
    D = 64 * H
    q16, k16, v16, do16 = bf(rnd(1, B, Lq, D)), bf(rnd(2, B, Lkv, D)), bf(rnd(3, B, Lkv, D)), bf(rnd(4, B, Lq, D))
    site, scale = ops.new_site(), 64 ** -0.5
    st = ops.rng.state("cuda")
    o = torch.empty(B * Lq, D, dtype=H16, device="cuda")
    lse = torch.empty(B * H * Lq, dtype=torch.float32, device="cuda")
    L.call("vpf_attention_fwd", q16, D, k16, D, v16, D, B, H, Lq, Lkv, 64, scale, p, st, site, o, D, lse)

    def run(flag):
        prev = L.debug_get("attn_ca_merged")
        L.debug_set("attn_ca_merged", flag)
        try:
            dq = torch.full((B * Lq, D), float("nan"), dtype=H16, device="cuda")
            dk = torch.full((B * Lkv, D), float("nan"), dtype=H16, device="cuda")
            dv = torch.full((B * Lkv, D), float("nan"), dtype=H16, device="cuda")
            L.call("vpf_attention_bwd", q16, D, k16, D, v16, D, o, D, do16, D, lse, B, H, Lq, Lkv, 64, scale, p, st, site,
                   dq, D, dk, D, dv, D, torch.empty(B * H * Lq, dtype=torch.float32, device="cuda"))
            torch.cuda.synchronize()
            return dq, dk, dv
        finally:
            L.debug_set("attn_ca_merged", prev)

    two, one, again = run(0), run(1), run(1)
    for name, a, b, c in zip(("dq", "dk", "dv"), two, one, again):
        assert torch.isfinite(a.float()).all() and float(a.float().abs().max()) > 0, name
        assert torch.equal(b, c), ("not reproducible", name)
        assert torch.equal(a, b), (name, rel(b.float(), a.float()))


@pytest.mark.parametrize("B,H,Lq,Lkv", [(2, 4, 196, 196), (2, 2, 96, 1024), (1, 2, 300, 520), (2, 4, 128, 128), (2, 4, 144, 144)])
def test_attention_dropout_32bit_group_index_equals_64bit_bitwise(B, H, Lq, Lkv):
    """The attention kernels index the dropout hash with 32-bit group numbers when B*H*Lq*Lkv < 2^32 (VPF_ATTN_RNG32, default on);
    the masks -- hence o, lse, dq, dk, dv -- must be bit-identical to the 64-bit indexing (which the exported mask uses)."""
    from vipformer_amd import _lib as L
    from vipformer_amd import ops
    D = 64 * H
    q16, k16, v16, do16 = bf(rnd(1, B, Lq, D)), bf(rnd(2, B, Lkv, D)), bf(rnd(3, B, Lkv, D)), bf(rnd(4, B, Lq, D))
    site, scale, p = ops.new_site(), 64 ** -0.5, 0.1
    st = ops.rng.state("cuda")

    def run(flag):
        L.debug_set("attn_rng32", flag)
        try:
            o = torch.empty(B * Lq, D, dtype=H16, device="cuda")
            lse = torch.empty(B * H * Lq, dtype=torch.float32, device="cuda")
            L.call("vpf_attention_fwd", q16, D, k16, D, v16, D, B, H, Lq, Lkv, 64, scale, p, st, site, o, D, lse)
            dq = torch.empty(B * Lq, D, dtype=H16, device="cuda")
            dk = torch.empty(B * Lkv, D, dtype=H16, device="cuda")
            dv = torch.empty(B * Lkv, D, dtype=H16, device="cuda")
            L.call("vpf_attention_bwd", q16, D, k16, D, v16, D, o, D, do16, D, lse, B, H, Lq, Lkv, 64, scale, p, st, site,
                   dq, D, dk, D, dv, D, torch.empty(B * H * Lq, dtype=torch.float32, device="cuda"))
            torch.cuda.synchronize()
            return o, lse, dq, dk, dv
        finally:
            L.debug_set("attn_rng32", 1)

    for name, a, b in zip(("o", "lse", "dq", "dk", "dv"), run(0), run(1)):
        assert torch.equal(a, b), name


@pytest.mark.parametrize("B,H,Lq,Lkv,p", [(3, 2, 40, 70, 0.0), (2, 2, 96, 1024, 0.1), (3, 4, 196, 196, 0.1), (2, 1, 33, 300, 0.5), (3, 2, 96, 96, 0.0)])
def test_attention_pad_mask_fwd_bwd(B, H, Lq, Lkv, p):
    """vpf_attention_fwd_pad / _bwd_pad (partseg.py:73-77: masked_fill_(pad_mask, -finfo.max) in front of the softmax) against torch
    fp32 on the same h16 operands and the kernel's own dropout mask: a tail mask, a scattered mask, and -- last batch row -- every
    key padded (uniform attention; dq = dk = 0 there, dv = mean of dout).  Without a mask set the result is the unmasked kernels'."""
    from vipformer_amd import _lib as L
    from vipformer_amd import ops
    D = 64 * H
    q, k, v, do = rnd(1, B, Lq, D), rnd(2, B, Lkv, D), rnd(3, B, Lkv, D), rnd(4, B, Lq, D)
    q16, k16, v16, do16 = bf(q), bf(k), bf(v), bf(do)
    g = torch.Generator().manual_seed(5)
    pad = torch.zeros(B, Lkv, dtype=torch.bool)
    pad[0, Lkv - Lkv // 3:] = True
    if B > 2:
        pad[1] = torch.rand(Lkv, generator=g) < 0.5
    pad[B - 1] = True
    pad8 = pad.to(torch.uint8).cuda()
    site = ops.new_site()
    scale = 64 ** -0.5
    st = ops.rng.state("cuda")
    o = torch.empty(B * Lq, D, dtype=H16, device="cuda")
    lse = torch.empty(B * H * Lq, dtype=torch.float32, device="cuda")
    L.call("vpf_attention_fwd_pad", q16, D, k16, D, v16, D, B, H, Lq, Lkv, 64, scale, p, st, site, o, D, lse, pad8)
    keep = ops.dropout_keep_mask(site, p, (B, H, Lq, Lkv), "cuda").float() if p > 0 else None
    split = lambda t, Lx: t.float().view(B, Lx, H, 64).permute(0, 2, 1, 3).contiguous().requires_grad_()
    qr, kr, vr = split(q16, Lq), split(k16, Lkv), split(v16, Lkv)
    s = torch.einsum("bhid,bhjd->bhij", qr, kr) * scale
    s = s.masked_fill(pad.cuda()[:, None, None, :], -torch.finfo(torch.float32).max)
    a = s.softmax(-1)
    if keep is not None:
        a = a * keep / (1 - p)
    oref = torch.einsum("bhij,bhjd->bhid", a, vr)
    og = o.float().view(B, Lq, H, 64).permute(0, 2, 1, 3)
    assert rel(og, oref) < 8e-3, f"fwd rel {rel(og, oref)}"
    lref = torch.logsumexp(s, -1)
    part = ~pad.all(dim=1)                                     # rows with a real key: the log-sum-exp is an ordinary number
    assert torch.allclose(lse.view(B, H, Lq)[part.cuda()], lref[part.cuda()], rtol=1e-4, atol=1e-4)
    assert bool((lse.view(B, H, Lq)[B - 1] < -1e37).all())     # every key padded: ~ -finfo.max, recognised by the backward kernels
    dq = torch.empty(B * Lq, D, dtype=H16, device="cuda")
    dk = torch.empty(B * Lkv, D, dtype=H16, device="cuda")
    dv = torch.empty(B * Lkv, D, dtype=H16, device="cuda")
    L.call("vpf_attention_bwd_pad", q16, D, k16, D, v16, D, o, D, do16, D, lse, B, H, Lq, Lkv, 64, scale, p, st, site,
           dq, D, dk, D, dv, D, torch.empty(B * H * Lq, dtype=torch.float32, device="cuda"), pad8)
    oref.backward(do16.float().view(B, Lq, H, 64).permute(0, 2, 1, 3))
    unsplit = lambda t, Lx: t.permute(0, 2, 1, 3).reshape(B * Lx, D)
    for name, got, ref in (("dq", dq, unsplit(qr.grad, Lq)), ("dk", dk, unsplit(kr.grad, Lkv)), ("dv", dv, unsplit(vr.grad, Lkv))):
        assert torch.isfinite(got.float()).all(), name
        r = rel(got.float(), ref)
        assert r < 1.5e-2, f"{name} rel {r}"
    # masked_fill_ overwrote the padded scores: no gradient through them
    assert float(dq.view(B, Lq, D)[B - 1].float().abs().max()) == 0.0 and float(dk.view(B, Lkv, D)[B - 1].float().abs().max()) == 0.0
    assert float(dk.view(B, Lkv, D)[0, Lkv - Lkv // 3:].float().abs().max()) == 0.0
    assert float(dv.view(B, Lkv, D)[0, Lkv - Lkv // 3:].float().abs().max()) == 0.0      # p = 0 beside real keys
    assert float(dv.view(B, Lkv, D)[B - 1].float().abs().max()) > 0.0                    # uniform attention still feeds v
    # a NULL mask is refused (the unmasked entry points exist for that), and an all-zero mask reproduces the unmasked kernels bit for bit
    o2 = torch.empty_like(o); lse2 = torch.empty_like(lse)
    with pytest.raises(L.VpfError):
        L.call("vpf_attention_fwd_pad", q16, D, k16, D, v16, D, B, H, Lq, Lkv, 64, scale, p, st, site, o2, D, lse2, None)
    L.call("vpf_attention_fwd_pad", q16, D, k16, D, v16, D, B, H, Lq, Lkv, 64, scale, p, st, site, o2, D, lse2, torch.zeros_like(pad8))
    o3 = torch.empty_like(o); lse3 = torch.empty_like(lse)
    ops_res = L.debug_get("attn_resident")
    L.debug_set("attn_resident", 0)                           # (the masked path never takes the LDS-resident self-attention kernels)
    try:
        L.call("vpf_attention_fwd", q16, D, k16, D, v16, D, B, H, Lq, Lkv, 64, scale, p, st, site, o3, D, lse3)
    finally:
        L.debug_set("attn_resident", ops_res)
    assert torch.equal(o2, o3) and torch.equal(lse2, lse3)


def test_attention_strided_qkv_views():
    """q/k/v as column slices of one [M,3D] buffer (how the fused QKV projection hands them over)."""
    from vipformer_amd import _lib as L
    from vipformer_amd import ops
    B, H, Lq = 2, 2, 96
    D = 128
    qkv = bf(rnd(1, B * Lq, 3 * D))
    o = torch.empty(B * Lq, D, dtype=H16, device="cuda")
    lse = torch.empty(B * H * Lq, dtype=torch.float32, device="cuda")
    L.call("vpf_attention_fwd", qkv, 3 * D, qkv[:, D:], 3 * D, qkv[:, 2 * D:], 3 * D, B, H, Lq, Lq, 64, 0.125, 0.0,
           ops.rng.state("cuda"), 1, o, D, lse)
    sp = lambda t: t.float().reshape(B, Lq, H, 64).permute(0, 2, 1, 3)
    oref, _ = attn_ref(sp(qkv[:, :D]), sp(qkv[:, D:2 * D]), sp(qkv[:, 2 * D:]), 0.125, None, 0.0)
    assert rel(o.float().view(B, Lq, H, 64).permute(0, 2, 1, 3), oref) < 8e-3


# ------------------------------------------------------------------------------------------ BN / pooling / small kernels
def test_batchnorm_pieces():
    from vipformer_amd import ops
    M, C = 4096, 256
    x = rnd(1, M, C, scale=2.0) + 0.3
    bn = torch.nn.BatchNorm1d(C).cuda()
    bn.weight.data = rnd(2, C) * 0.2 + 1; bn.bias.data = rnd(3, C) * 0.1
    ref = torch.nn.BatchNorm1d(C).cuda()
    ref.load_state_dict(bn.state_dict())
    x16 = bf(x)
    stat = ops._bn_stat(x16, C, bn, True)
    y = ops._bn_act(x16, C, stat, bn, True, False)
    xr = x16.float().requires_grad_()
    yr = torch.relu(ref(xr))
    assert rel(y, yr) < 1e-4
    assert torch.allclose(bn.running_mean, ref.running_mean, atol=1e-5) and torch.allclose(bn.running_var, ref.running_var, rtol=1e-4)
    assert int(bn.num_batches_tracked) == 1
    dy = bf(rnd(4, M, C))
    dx = ops._bn_bwd(dy, x16, C, stat, bn, True, True, False)
    yr.backward(dy.float())
    assert rel(dx, xr.grad) < 2e-4 and rel(bn.weight.grad, ref.weight.grad) < 2e-4 and rel(bn.bias.grad, ref.bias.grad) < 2e-4
    # eval mode uses the running statistics
    bn.eval(); ref.eval()
    se = ops._bn_stat(x16, C, bn, False)
    assert rel(ops._bn_act(x16, C, se, bn, False, False), ref(x16.float())) < 1e-5


@pytest.mark.parametrize("M,C", [(128, 256), (37, 64), (1, 64), (256, 512)])
def test_batchnorm_small_batch_one_kernel(M, C):
    """vpf_bn_small_fwd / vpf_bn_small_bwd (projection heads, partseg.py:519-525: BatchNorm1d + ReLU in training mode on a
    batch of 64 .. 256 rows as ONE kernel each way) against torch.nn.BatchNorm1d fp32.  Tolerances: the output is h16
    (2^-8 relative per element -> 3e-3 on the norm), statistics and gradients are fp32 sums (1e-4)."""
    from vipformer_amd import _lib as L
    x = rnd(11, M, C, scale=2.0) + 0.3
    bn = torch.nn.BatchNorm1d(C).cuda()
    bn.weight.data = rnd(12, C) * 0.2 + 1; bn.bias.data = rnd(13, C) * 0.1
    ref = torch.nn.BatchNorm1d(C).cuda()
    ref.load_state_dict(bn.state_dict())
    stat = torch.empty(2 * C, device="cuda"); y = torch.empty(M, C, dtype=H16, device="cuda")
    L.call("vpf_bn_small_fwd", x, M, C, bn.weight.data, bn.bias.data, float(bn.eps), float(bn.momentum), bn.running_mean, bn.running_var,
           bn.num_batches_tracked, stat, y, 1)
    xr = x.clone().requires_grad_()
    if M > 1:
        yr = torch.relu(ref(xr))
        assert rel(y.float(), yr) < 3e-3
        assert torch.allclose(bn.running_mean, ref.running_mean, atol=1e-5) and torch.allclose(bn.running_var, ref.running_var, rtol=1e-4, atol=1e-6)
    else:       # torch refuses a single row in training mode; the kernel follows the formula (var = 0 -> y = relu(beta))
        assert torch.allclose(y.float()[0], torch.relu(bn.bias.data).to(H16).float(), atol=1e-6)
        return
    assert int(bn.num_batches_tracked) == 1
    assert torch.allclose(stat[:C], x.mean(0), atol=1e-5)
    dy = rnd(14, M, C)
    for out_h16 in (0, 1):
        dx = torch.empty(M, C, dtype=H16 if out_h16 else torch.float32, device="cuda")
        dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
        L.call("vpf_bn_small_bwd", dy, x, stat, bn.weight.data, bn.bias.data, M, C, 1, dx, out_h16, dg, db)
        if out_h16 == 0:
            yr.backward(dy)
        assert rel(dx.float(), xr.grad) < (3e-3 if out_h16 else 2e-4)
        assert rel(dg, ref.weight.grad) < 2e-4 and rel(db, ref.bias.grad) < 2e-4
    # shapes the kernel does not take are refused loudly
    with pytest.raises(RuntimeError):
        L.call("vpf_bn_small_fwd", x, M, C - 1, bn.weight.data, bn.bias.data, 1e-5, 0.1, None, None, None, stat, y, 1)


def test_group_max_concat_pool():
    from vipformer_amd import _lib as L
    NG, K, C = 50, 32, 128
    h = bf(rnd(1, NG * K, C))
    out = torch.empty(NG, C, dtype=torch.float32, device="cuda"); arg = torch.empty(NG, C, dtype=torch.uint8, device="cuda")
    L.call("vpf_group_max_fwd", h, NG, K, C, out, 0, arg)
    mv, mi = h.float().view(NG, K, C).max(1)
    assert torch.equal(out, mv)
    assert torch.equal(torch.gather(h.float().view(NG, K, C), 1, arg.long().unsqueeze(1)).squeeze(1), mv)
    dout = rnd(2, NG, C)
    dh = torch.empty(NG * K, C, dtype=H16, device="cuda")
    L.call("vpf_group_max_bwd", dout, 0, arg, NG, K, C, dh)
    ref = torch.zeros(NG, K, C, device="cuda").scatter_(1, arg.long().unsqueeze(1), bf(dout).float().unsqueeze(1))
    assert torch.equal(dh.float().view(NG, K, C), ref)
    g16 = bf(out)
    feat = torch.empty(NG * K, 2 * C, dtype=H16, device="cuda")
    L.call("vpf_g2e_concat_fwd", g16, h, NG * K, K, C, feat)
    assert torch.equal(feat.view(NG, K, 2 * C)[:, :, :C], g16.unsqueeze(1).expand(NG, K, C)) and torch.equal(feat[:, C:], h)
    dfeat = bf(rnd(3, NG * K, 2 * C))
    dh2 = torch.empty(NG * K, C, dtype=H16, device="cuda")
    L.call("vpf_g2e_concat_bwd", dfeat, arg, NG, K, C, dh2)
    dfv = dfeat.float().view(NG, K, 2 * C)
    ref2 = dfv[:, :, C:] + torch.zeros(NG, K, C, device="cuda").scatter_(1, arg.long().unsqueeze(1), dfv[:, :, :C].sum(1, keepdim=True))
    assert rel(dh2.float().view(NG, K, C), ref2) < 4e-3
    # token pooling
    from vipformer_amd import ops
    x = rnd(4, 6, 96, 64).requires_grad_()
    o = ops.PoolFn.apply(x)
    r = torch.cat([x.max(1)[0], x.mean(1)], 1)
    assert torch.allclose(o, r, atol=1e-6)
    gsel = rnd(5, 6, 128)
    (gx,) = torch.autograd.grad((o * gsel).sum(), x)
    (gr,) = torch.autograd.grad((r * gsel).sum(), x)
    assert torch.allclose(gx, gr, atol=1e-6)


def test_ntxent_fwd_bwd_vs_oracle():
    from oracle import torch_oracle as O
    from vipformer_amd import ops
    for b, D in ((64, 256), (4, 64), (33, 384)):
        z0, z1 = rnd(1, b, D).requires_grad_(), rnd(2, b, D).requires_grad_()
        loss = ops.ntxent_loss(z0, z1, 0.1)
        (loss * 1.7).backward()
        c0, c1 = z0.detach().cpu().requires_grad_(), z1.detach().cpu().requires_grad_()
        lr = O.ntxent(c0, c1, 0.1)
        (lr * 1.7).backward()
        assert abs(loss.item() - lr.item()) < 1e-4 * max(1, abs(lr.item()))
        assert rel(z0.grad.cpu(), c0.grad) < 1e-4 and rel(z1.grad.cpu(), c1.grad) < 1e-4


def test_adamw_matches_torch():
    from vipformer_amd import _lib as L
    n = 100003
    p0, g = rnd(1, n), rnd(2, n, scale=0.01)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3)
    p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    sh = torch.empty(n, dtype=H16, device="cuda")
    hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 0.01, 1.0, 0.0, 0.0] + [0.0] * 8, device="cuda")        # no loss scale
    for it in range(3):
        gi = g * (it + 1)
        ref.grad = gi.clone(); opt.step()
        L.call("vpf_adamw_step", p, gi, m, v, sh, n, hyper, 1)
    assert torch.allclose(p, ref.data, rtol=1e-5, atol=1e-7) and hyper[6].item() == 3.0
    assert torch.equal(sh, p.to(H16))


def test_adamw_follows_gradscaler_step_and_update():
    """torch.cuda.amp.GradScaler's step / update on the device (pretrain.py:154,209-211; include/vipformer_hip.h vpf_adamw_step):
    gradients carry the loss scale and are divided by it; vpf_grad_check finds an inf / NaN anywhere in the flat gradient, that
    step changes nothing (parameters, moments, bias-correction counter) and halves the scale; `growth interval` good steps in a
    row double it; GradScaler.unscale_ (hyper[15]) hands AdamW gradients that are already unscaled."""
    from vipformer_amd import _lib as L
    n = 100003
    S = 1024.0
    p0, g = rnd(1, n), rnd(2, n, scale=0.01)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3)
    scaler = torch.amp.GradScaler("cuda", init_scale=S, growth_interval=3)
    scaler.scale(torch.zeros(1, device="cuda"))            # (GradScaler creates its device-side scale lazily, at the first scale())
    p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    sh = torch.empty(n, dtype=H16, device="cuda")
    #                      lr    b1   b2     eps   wd   gs   step skip  S  tracker interval found growth backoff skipped unscaled
    hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 0.01, 1.0, 0.0, 0.0, S, 0.0, 3.0, 0.0, 2.0, 0.5, 0.0, 0.0], device="cuda")
    bad_at = {2: float("inf"), 5: float("nan")}
    for it in range(9):
        scale_now = hyper[8].item()
        assert scale_now == scaler.get_scale(), (it, scale_now, scaler.get_scale())
        gi = g * (it + 1) * scale_now                      # what backward leaves behind scaler.scale(loss)
        if it in bad_at:
            gi[n - 2] = bad_at[it]                         # (the last float4 group's tail: the check's remainder path)
        ref.grad = gi.clone()
        scaler.step(opt); scaler.update()
        gk = gi.clone()
        if it == 7:                                        # GradScaler.unscale_ before the step (gradient clipping would sit here)
            gk /= scale_now; hyper[15] = 1.0
        L.call("vpf_grad_check", gk, n, hyper)
        assert hyper[11].item() == (1.0 if it in bad_at else 0.0)
        L.call("vpf_adamw_step", p, gk, m, v, sh, n, hyper, 1)
        assert hyper[11].item() == 0.0 and hyper[15].item() == 0.0
        assert torch.allclose(p, ref.data, rtol=1e-5, atol=1e-7), it
    assert hyper[6].item() == 7.0 and hyper[14].item() == 2.0          # 9 steps, 2 skipped
    assert torch.equal(sh, p.to(H16))
    # a clean flat gradient of a length that is not a multiple of 4, and one whose only bad value sits in the remainder
    for bad in (False, True):
        x = torch.ones(4099, device="cuda")[:4098 + 0]
        x = x[:4098].clone()
        if bad:
            x[4097] = float("-inf")
        hyper[11] = 0.0
        L.call("vpf_grad_check", x, x.numel(), hyper)
        assert hyper[11].item() == float(bad)
    hyper[11] = 0.0


def test_patchify_on_permuted_nchw_view():
    from oracle import torch_oracle as O
    from vipformer_amd import _lib as L
    imgs = Hh.synth_images(3, 2, 32, 48).cuda()                     # [B,H,W,3] view of NCHW
    assert not imgs.is_contiguous()
    out = torch.empty(2 * (32 // 8) * (48 // 8), 8 * 8 * 3, dtype=H16, device="cuda")
    sb, sh, sw, sc = imgs.stride()
    L.call("vpf_patchify", imgs, sb, sh, sw, sc, 2, 32, 48, 3, 8, out)
    assert torch.equal(out.view(2, -1, 192), bf(O.patchify(imgs.cpu(), 8)).cuda())


def test_fused_pretrain_losses_match_the_two_ntxent_calls():
    """vpf_pretrain_loss_fwd / _bwd (both NT-Xent losses, the view mean and the weighted sum) against the per-loss kernels
    composed with torch ops: same values and gradients (fp32, tolerance 1e-5 relative)."""
    from vipformer_amd import ops
    torch.manual_seed(0)
    b, D, w = 24, 256, 0.7
    f = torch.randn(2 * b, D, device="cuda", requires_grad=True)
    g = torch.randn(b, D, device="cuda", requires_grad=True)
    total, parts = ops.pretrain_losses(f, g, 0.1, w)
    total.backward()
    df, dg = f.grad.clone(), g.grad.clone()
    f.grad = None; g.grad = None
    f1, f2 = f[:b], f[b:]
    li = ops.ntxent_loss(f1, f2, 0.1)
    lc = ops.ntxent_loss((f1 + f2) / 2, g, 0.1)
    ref = li + w * lc
    ref.backward()
    total, ref, li, lc = total.detach(), ref.detach(), li.detach(), lc.detach()
    assert abs(float(total) - float(ref)) < 1e-5 * abs(float(ref))
    assert abs(float(parts[0]) - float(li)) < 1e-5 * abs(float(li)) and abs(float(parts[1]) - float(lc)) < 1e-5 * abs(float(lc))
    assert (df - f.grad).abs().max().item() < 1e-5 * f.grad.abs().max().item() + 1e-9
    assert (dg - g.grad).abs().max().item() < 1e-5 * g.grad.abs().max().item() + 1e-9


def test_grouped_wgrad_workspace_split_k():
    """vpf_wgrad_group with the split-K workspace (partial tiles + last-arriver reduction, no atomics on dW): the four weight
    gradients of an encoder layer at the benchmark's token count and at ragged sizes, accumulated INTO non-zero buffers, launched
    twice in a row (the arrival counters must come back to zero), against fp32 matmuls of the same h16 operands."""
    from vipformer_amd import ops
    ops.cfg.wgrad_deterministic = True
    try:
        _grouped_wgrad_cases(ops)
        ws = ops.wgrad_workspace("cuda")
        assert int(ws[:1024].view(torch.int32).abs().sum()) == 0          # counters are back to zero
    finally:
        ops.cfg.wgrad_deterministic = False
    _grouped_wgrad_cases(ops)                                             # the default: fp32 atomics
    from vipformer_amd import _lib
    keep, keep_dma = _lib.debug_get("wgroup_cfg"), _lib.debug_get("wgroup_dma")
    try:
        _lib.debug_set("wgroup_dma", 0)                                   # (the register-staged kernels: the LDS-DMA one has its own test)
        for cfg in (2, 8):                                                # 8: two K slices per 8-wave workgroup, LDS exchange, half the atomics
            _lib.debug_set("wgroup_cfg", cfg)
            _grouped_wgrad_cases(ops)
    finally:
        _lib.debug_set("wgroup_cfg", keep); _lib.debug_set("wgroup_dma", keep_dma)


def test_grouped_wgrad_lds_dma_kernel():
    """gemm_wgrad_dma_kernel (round 5: 256 x 128 tiles, stages by LDS-DMA, three deep) against fp32 matmuls of the same h16 operands: the
    four weight gradients of an encoder layer at the point-cloud and image token counts (12 288 = 192 x 64, 12 544 = 196 x 64 tokens:
    two K slices of 96 and 98 stages) and a whole stack of 28 problems, accumulated INTO non-zero buffers, bias sums on and off, twice
    in a row; a launch with ONE non-conforming problem must take the register-staged kernel and still be right."""
    from vipformer_amd import _lib, ops
    keep, keep_tn, keep_wgs = (_lib.debug_get(k) for k in ("wgroup_dma", "wgroup_dma_tn", "wgroup_wgs"))
    layer = [(256, 512), (512, 256), (256, 256), (768, 256)]
    stack = [(256, 512), (512, 256), (256, 256), (256, 256)] + layer * 6
    try:
        _lib.debug_set("wgroup_dma", 1)
        # (tile columns, tokens, problems, workgroup target): 256 x 256 tiles (four stages of 32 tokens) where every K_in allows it, 256 x 128
        # (three stages of 64) forced and for a group that holds a 128-column problem
        for tn, M, shapes, wgs in ((0, 12288, layer, 0), (128, 12288, layer, 0), (0, 12544, layer, 0), (128, 12544, layer, 0),
                                   (0, 12288, stack, 0), (128, 12288, stack, 0), (128, 12544, stack, 0), (0, 320, layer, 0), (128, 320, layer, 0),
                                   (0, 4096, layer + [(256, 128)], 0), (0, 12288, layer + [(64, 128)], 0),
                                   # three slices (86, 86, 84 stages) per tile: XCD lists of ~40 tile-slices, two rounds of workgroups
                                   (128, 16384, stack, 384),
                                   # D = 384 / hidden 1536 (config 4): N_out multiples of 128 only -> the 128 x 128 configuration
                                   (0, 4096, [(384, 1536), (1536, 384), (384, 384), (1152, 384)], 0), (0, 3136, [(384, 1536), (1536, 384), (384, 384), (1152, 384)], 0),
                                   (0, 12288, [(384, 1536), (1536, 384), (384, 384), (1152, 384)] * 3, 0)):
            _lib.debug_set("wgroup_dma_tn", tn); _lib.debug_set("wgroup_wgs", wgs)
            jobs = []
            for i, (N, K) in enumerate(shapes):
                dy, x = bf(rnd(100 + i, M, N)), bf(rnd(200 + i, M, K))
                jobs.append((dy, x, N, K, (torch.ones(N, device="cuda") if i % 4 != 3 else None)))
            outs = [(torch.ones(N, K, device="cuda"), (torch.ones(N, device="cuda") if db is not None else None)) for dy, x, N, K, db in jobs]
            for rep in range(2):
                wg = ops.WgradBatch(cap=ops.WgradBatch.CAP)
                for (dy, x, N, K, _), (dW, db) in zip(jobs, outs):
                    wg.add(dy, x, N, K, dW, db)
                wg.flush()
            torch.cuda.synchronize()
            for (dy, x, N, K, _), (dW, db) in zip(jobs, outs):
                ref = dy.float().t() @ x.float()
                assert rel(dW, 1.0 + 2.0 * ref) < 2e-5, (M, N, K, rel(dW, 1.0 + 2.0 * ref))
                if db is not None:
                    assert rel(db, 1.0 + 2.0 * dy.float().sum(0)) < 2e-5, (M, N, K)
    finally:
        for k, v in (("wgroup_dma", keep), ("wgroup_dma_tn", keep_tn), ("wgroup_wgs", keep_wgs)):
            _lib.debug_set(k, v)


def _grouped_wgrad_cases(ops):
    for M, shapes in ((12288, [(256, 512), (512, 256), (256, 256), (768, 256)]), (1000, [(64, 128), (136, 72)]), (40, [(8, 8)])):
        jobs = []
        for i, (N, K) in enumerate(shapes):
            dy, x = bf(rnd(10 + i, M, N)), bf(rnd(20 + i, M, K))
            jobs.append((dy, x, N, K, torch.ones(N, K, device="cuda"), torch.ones(N, device="cuda")))
        for rep in range(2):
            wg = ops.WgradBatch()
            for dy, x, N, K, dW, db in jobs:
                wg.add(dy, x, N, K, dW, db)
            wg.flush()
        for dy, x, N, K, dW, db in jobs:
            ref = dy.float().t() @ x.float()
            assert rel(dW, 1.0 + 2.0 * ref) < 2e-5, (M, N, K, rel(dW, 1.0 + 2.0 * ref))
            assert rel(db, 1.0 + 2.0 * dy.float().sum(0)) < 2e-5


def test_timeline_marks_are_ordered_and_capturable():
    """ops.Timeline (vpf_stamp): marks on a stream come back in issue order, microseconds apart by at least the work between them, and a
    mark captured into a hipGraph is refreshed by every replay."""
    from vipformer_amd import ops
    tl = ops.Timeline(torch.device("cuda", 0))
    x = torch.randn(4096, 4096, device="cuda")
    tl.mark("a")
    y = x @ x
    tl.mark("b")
    torch.cuda.synchronize()
    t = tl.read()
    assert t["a"] == 0.0 and 1.0 < t["b"] < 1e6, t
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        tl.mark("c")                                    # (warm-up of the launch path outside capture)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        tl.mark("c")
    g.replay(); torch.cuda.synchronize()
    c1 = int(tl.buf[tl.names.index("c")])
    g.replay(); torch.cuda.synchronize()
    c2 = int(tl.buf[tl.names.index("c")])
    assert c2 > c1 > 0 and float(y[0, 0]) == float(y[0, 0])


# ------------------------------------------------------------------------------------------ encoder row-block kernels, round 3
def _tail_case(B, Lq, with_next, with_pos, seed, training=True):
    """A closure that runs one vpf_sa_layer_fwd launch (attention_done) at D = 256 on seeded operands -- the SAME modules (dropout
    sites), weights and dropout state every call -- and returns every tensor the kernel writes."""
    import torch.nn as nn
    from vipformer_amd import ops
    from vipformer_amd.model.pointcloud.partseg import SelfAttentionLayer
    D, H = 256, 4
    M = B * Lq
    torch.manual_seed(seed)
    layers = nn.ModuleList([SelfAttentionLayer(H, D, 2, 0.0, 0.1, 0.5) for _ in range(2)]).cuda()
    layers.train(training)
    blocks = [(l[0].module.attention, l[1].module, True, True) for l in layers]
    packed = ops._pack_blocks(blocks, layers[0], "cuda")
    st = ops.rng.state("cuda")
    base = rnd(seed + 1, M, D); pos = rnd(seed + 2, Lq, D) if with_pos else None
    o = bf(rnd(seed + 3, M, D)); lse = torch.zeros(B * H * Lq, device="cuda")
    att, mlp = layers[0][0].module.attention, layers[0][1].module
    nxt = (layers[1][0].module.norm, packed[1]["Wqkv"]) if with_next else None

    def run():
        with ops.rng.pinned():
            saved, out, head = ops._tail_fwd(att, mlp, layers[0][0], layers[0][1], packed[0], training, st, B, Lq, o, base, o, lse, nxt, pos,
                                             Lq if with_pos else 0, "cuda")
        torch.cuda.synchronize()
        return list(saved) + [out] + (list(head) if head is not None else [])
    return run


@pytest.mark.parametrize("B,Lq,with_next,with_pos", [(128, 96, True, True), (3, 50, True, True), (2, 196, False, False), (1, 7, True, False)])
def test_sa_rows_fwd_equals_the_one_per_cu_kernel_bitwise(B, Lq, with_next, with_pos):
    """sa_rows_fwd_kernel (two workgroups per CU: residual in registers, wave-private transposition slices) performs the SAME
    arithmetic in the same order as sa_layer_fwd_kernel<.., false, 1> (one per CU, f32 residual tile in LDS): every output -- x1,
    LayerNorm statistics, n2, u, h, out (+ pos), next n1 and q|k|v -- must be bit-identical, dropout on, whole and ragged blocks."""
    from vipformer_amd import _lib
    run = _tail_case(B, Lq, with_next, with_pos, 40)
    outs = []
    # (sa_wg2, sa_rb, sa_stagger): the one-per-CU kernel; the round-3 geometry twice; round 5: 16 waves in lockstep, the same 16 waves
    # as two DECOUPLED 8-wave groups (LDS-counter barriers per group), and those with the second group started 20 x 64 cycles late
    # 4th field: the cache policy of the row stores (0 plain, 1 sc1 write-through, 2 nt, 3 sc0 sc1) -- the bytes are the same
    variants = [(0, 0, 0, 0), (3, 0, 0, 0), (3, 0, 0, 0), (1, 12, 0, 0), (1, 13, 0, 0), (1, 13, 20, 0), (1, 13, 20, 1), (1, 13, 0, 2), (1, 13, 0, 3)]
    keys = ("sa_wg2", "sa_rb", "sa_stagger", "sa_store")
    before = {k: _lib.debug_get(k) for k in keys}          # (a suite run with VPF_SA_RB / VPF_SA_WG2 set keeps ITS geometry afterwards)
    for wg2, rb, stg, pol in variants:
        _lib.debug_set("sa_wg2", wg2); _lib.debug_set("sa_rb", rb); _lib.debug_set("sa_stagger", stg); _lib.debug_set("sa_store", pol)
        try:
            outs.append(run())
        finally:
            for k in keys:
                _lib.debug_set(k, int(before[k]))
    names = ["x1", "mean2", "rstd2", "n2", "u", "h", "out", "mean1n", "rstd1n", "n1n", "qkv_next"]
    assert len(outs[0]) == len(outs[1])
    for n, p, q, r in zip(names, outs[0], outs[1], outs[2]):
        assert torch.equal(q, r), ("not reproducible", n)
        assert torch.equal(p, q), (n, (p.float() - q.float()).abs().max().item(), (p != q).float().mean().item())
    for v, o in zip(variants[3:], outs[3:]):
        for n, p, q in zip(names, outs[0], o):
            assert torch.equal(p, q), (v, n, (p.float() - q.float()).abs().max().item(), (p != q).float().mean().item())


@pytest.mark.parametrize("M,with_dsum,dsum_init", [(12288, True, 1), (150, True, 0), (392, False, 0)])
def test_sa_rows_bwd_equals_the_one_per_cu_kernels_bitwise(M, with_dsum, dsum_init):
    """sa_rows_bwd_mlp_kernel / sa_rows_bwd_qkv_kernel against sa_bwd_mlp_rows_kernel / sa_bwd_qkv_rows_kernel at D = 256 on the same
    operands, dropout on: dz2, du, dx1, dz1, do, dbase and the running positional-gradient sum must be bit-identical (same arithmetic,
    same order); the LayerNorm parameter-gradient partials are grouped per 32 instead of 64 tokens, so their fold agrees to fp32
    summation order."""
    import torch.nn as nn
    from vipformer_amd import _lib, ops
    from vipformer_amd.model.pointcloud.partseg import SelfAttentionLayer
    D, Hd = 256, 512
    torch.manual_seed(3)
    layer = SelfAttentionLayer(4, D, 2, 0.0, 0.1, 0.5).cuda()
    layer.train()
    pk = ops._pack_blocks([(layer[0].module.attention, layer[1].module, True, True)], layer, "cuda")[0]
    st = ops.rng.state("cuda")
    d = rnd(1, M, D); u = bf(rnd(2, M, Hd)); x1 = rnd(3, M, D); base = rnd(4, M, D); dqkv = bf(rnd(5, M, 3 * D, scale=0.1))
    m2 = x1.mean(1).contiguous(); r2 = (x1.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    m1 = base.mean(1).contiguous(); r1 = (base.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    dsum0 = rnd(6, M, D)
    ln1, ln2 = layer[0].module.norm, layer[1].module[0]
    with torch.no_grad():
        ln1.weight.copy_(rnd(7, D) * 0.2 + 1.0); ln2.weight.copy_(rnd(8, D) * 0.2 + 1.0)

    def run():
        nwg = ops.pgrad_rows(M, D)
        out = dict(dz2=torch.empty(M, D, dtype=H16, device="cuda"), du=torch.empty(M, Hd, dtype=H16, device="cuda"),
                   dx1=torch.empty(M, D, device="cuda"), dz1=torch.empty(M, D, dtype=H16, device="cuda"),
                   do=torch.empty(M, D, dtype=H16, device="cuda"), dbase=torch.empty(M, D, device="cuda"), dsum=dsum0.clone())
        pg = torch.zeros(2, nwg * 2 * D, device="cuda")
        a = _lib.SaLayerBwd()
        a.M, a.D, a.hidden, a.rng = M, D, Hd, st.data_ptr()
        a.p_res1, a.site_res1, a.p_res2, a.site_res2 = 0.5, layer[0].site, 0.5, layer[1].site
        a.d, a.u, a.x1, a.mean2, a.rstd2, a.ln2_g = d.data_ptr(), u.data_ptr(), x1.data_ptr(), m2.data_ptr(), r2.data_ptr(), ln2.weight.data.data_ptr()
        a.W2T, a.W1T, a.WoT = pk["W2T"].data_ptr(), pk["W1T"].data_ptr(), pk["WoT"].data_ptr()
        a.dz2, a.du, a.dx1, a.dz1, a.dout_attn = (out[k].data_ptr() for k in ("dz2", "du", "dx1", "dz1", "do"))
        a.pgrad2, a.pgrad1 = pg[1].data_ptr(), pg[0].data_ptr()
        a.dqkv, a.WqkvT, a.base, a.mean1, a.rstd1, a.ln1_g = dqkv.data_ptr(), pk["WqkvT"].data_ptr(), base.data_ptr(), m1.data_ptr(), r1.data_ptr(), ln1.weight.data.data_ptr()
        a.dbase, a.dsum, a.dsum_init = out["dbase"].data_ptr(), (out["dsum"].data_ptr() if with_dsum else None), dsum_init
        _lib.call_struct("vpf_sa_layer_bwd_mlp", a)
        _lib.call_struct("vpf_sa_layer_bwd_qkv", a)
        torch.cuda.synchronize()
        out["pg2"] = pg[1].view(nwg, 2 * D).sum(0); out["pg1"] = pg[0].view(nwg, 2 * D).sum(0)
        return out

    res = []
    for wg2 in (0, 3, 3):
        _lib.debug_set("sa_wg2", wg2)
        try:
            res.append(run())
        finally:
            _lib.debug_set("sa_wg2", 0)
    for k in ("dz2", "du", "dx1", "dz1", "do", "dbase", "dsum"):
        assert torch.equal(res[1][k], res[2][k]), ("not reproducible", k)
        assert torch.equal(res[0][k], res[1][k]), (k, (res[0][k].float() - res[1][k].float()).abs().max().item())
    for k in ("pg1", "pg2"):
        assert rel(res[1][k], res[0][k]) < 1e-5, (k, rel(res[1][k], res[0][k]))


@pytest.mark.parametrize("M", [12288, 200])
def test_sa_bwd_qkv_mlp_one_launch_equals_two_bitwise(M):
    """vpf_sa_layer_bwd_qkv_mlp (the qkv half of a layer and the MLP half of the layer below in one workgroup, the gradient rows handed
    over through LDS) against the two launches it replaces (VPF_SA_BWD_FUSE = 0), dropout on, ragged last block: every output of both
    halves and both LayerNorms' parameter-gradient partials must be bit-identical."""
    import ctypes
    import torch.nn as nn
    from vipformer_amd import _lib, ops
    from vipformer_amd.model.pointcloud.partseg import SelfAttentionLayer
    D, Hd = 256, 512
    torch.manual_seed(3)
    layers = nn.ModuleList([SelfAttentionLayer(4, D, 2, 0.0, 0.1, 0.5) for _ in range(2)]).cuda()
    layers.train()
    pks = ops._pack_blocks([(l[0].module.attention, l[1].module, True, True) for l in layers], layers[0], "cuda")
    st = ops.rng.state("cuda")
    u = bf(rnd(2, M, Hd)); x1 = rnd(3, M, D); base = rnd(4, M, D); dqkv = bf(rnd(5, M, 3 * D, scale=0.1)); dx1_up = rnd(9, M, D)
    m2 = x1.mean(1).contiguous(); r2 = (x1.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    m1 = base.mean(1).contiguous(); r1 = (base.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    dsum0 = rnd(6, M, D)
    up, low = layers[1], layers[0]
    nwg = ops.pgrad_rows(M, D)

    def run(fuse):
        out = dict(dbase=torch.empty(M, D, device="cuda"), dsum=dsum0.clone(),
                   dz2=torch.empty(M, D, dtype=H16, device="cuda"), du=torch.empty(M, Hd, dtype=H16, device="cuda"),
                   dx1=torch.empty(M, D, device="cuda"), dz1=torch.empty(M, D, dtype=H16, device="cuda"),
                   do=torch.empty(M, D, dtype=H16, device="cuda"))
        pg = torch.zeros(2, nwg * 2 * D, device="cuda")
        a = _lib.SaLayerBwd()                                   # qkv half of the upper layer
        a.M, a.D, a.hidden, a.rng = M, D, Hd, st.data_ptr()
        a.dqkv, a.WqkvT, a.base, a.mean1, a.rstd1, a.ln1_g = (dqkv.data_ptr(), pks[1]["WqkvT"].data_ptr(), base.data_ptr(), m1.data_ptr(), r1.data_ptr(),
                                                             up[0].module.norm.weight.data.data_ptr())
        a.dx1, a.dbase, a.dsum, a.dsum_init, a.pgrad1 = dx1_up.data_ptr(), out["dbase"].data_ptr(), out["dsum"].data_ptr(), 0, pg[0].data_ptr()
        b = _lib.SaLayerBwd()                                   # MLP half of the layer below: its d is the dbase above
        b.M, b.D, b.hidden, b.rng = M, D, Hd, st.data_ptr()
        b.p_res1, b.site_res1, b.p_res2, b.site_res2 = 0.5, low[0].site, 0.5, low[1].site
        b.d, b.u, b.x1, b.mean2, b.rstd2, b.ln2_g = out["dbase"].data_ptr(), u.data_ptr(), x1.data_ptr(), m2.data_ptr(), r2.data_ptr(), low[1].module[0].weight.data.data_ptr()
        b.W2T, b.W1T, b.WoT = pks[0]["W2T"].data_ptr(), pks[0]["W1T"].data_ptr(), pks[0]["WoT"].data_ptr()
        b.dz2, b.du, b.dx1, b.dz1, b.dout_attn = (out[k].data_ptr() for k in ("dz2", "du", "dx1", "dz1", "do"))
        b.pgrad2 = pg[1].data_ptr()
        _lib.debug_set("sa_bwd_fuse", fuse)
        try:
            _lib.call_struct("vpf_sa_layer_bwd_qkv_mlp", a, ctypes.addressof(b))
            torch.cuda.synchronize()
        finally:
            _lib.debug_set("sa_bwd_fuse", 1)
        out["pg"] = pg
        return out

    two, one, again = run(0), run(1), run(1)
    assert float(two["dx1"].abs().max()) > 0 and float(two["do"].float().abs().max()) > 0
    for k in two:
        assert torch.equal(one[k], again[k]), ("not reproducible", k)
        assert torch.equal(two[k], one[k]), (k, rel(one[k], two[k]))


@pytest.mark.parametrize("B,N", [(8, 1024), (3, 333)])
def test_adapter_kv_bwd_rows_kernel_equals_the_round2_kernel_bitwise(B, N):
    """adapter_kv_bwd_rows_kernel (sa_rows.hip: two workgroups per CU) against adapter_kv_bwd_kernel (sa_layer.hip, VPF_SA_WG2 bit 2)
    on the operands a real AdapterKVFn forward saved (ragged last block included): dxkv, da1 and the kv LayerNorm's
    parameter-gradient partial rows must be BIT-identical -- same arithmetic, same rounding points, same 64-token grouping."""
    from vipformer_amd import _lib, ops
    from vipformer_amd.model.pointcloud import PointCloudInputAdapter
    from vipformer_amd.model.pointcloud.partseg import CrossAttention
    D = 256
    torch.manual_seed(11)
    adapter = PointCloudInputAdapter((N, 3), D).cuda()
    cross = CrossAttention(4, D, D, D, 0.1).cuda()
    with torch.no_grad():
        cross.kv_norm.weight.copy_(rnd(7, D) * 0.2 + 1.0)
    params = list(adapter.parameters()) + list(cross.kv_norm.parameters()) + [cross.attention.k_proj.weight, cross.attention.v_proj.weight]
    ops.clear_managed_shadows()
    kv = ops.AdapterKVFn.apply(rnd(1, B, N, 3) * 0.5, adapter, cross, *params)
    x, a1, xkv, mk, rk, nk = kv.grad_fn.saved_tensors
    pk = adapter._vpf_packed_kv
    M, n2, nkv = B * N, D * 64, 2 * D * D
    nwg = (M + 63) // 64
    dkv = bf(rnd(2, M, 2 * D, scale=0.1))

    def run(bit):
        out = dict(dxkv=torch.zeros(M, D, dtype=H16, device="cuda"), da1=torch.zeros(M, 64, dtype=H16, device="cuda"),
                   pg=torch.zeros(nwg, 2 * D, device="cuda"))
        a = _lib.AdapterKvBwd()
        a.M, a.C, a.D = M, 3, D
        a.dkv, a.WkvT, a.xkv, a.mean, a.rstd, a.lnkv_g = dkv.data_ptr(), pk[2 * n2 + nkv:].data_ptr(), xkv.data_ptr(), mk.data_ptr(), rk.data_ptr(), cross.kv_norm.weight.data.data_ptr()
        a.W2T = pk[n2 + nkv:].data_ptr()
        a.dxkv, a.da1, a.pgrad_kv = out["dxkv"].data_ptr(), out["da1"].data_ptr(), out["pg"].data_ptr()
        _lib.debug_set("sa_wg2", bit)
        try:
            _lib.call_struct("vpf_adapter_kv_bwd", a)
            torch.cuda.synchronize()
        finally:
            _lib.debug_set("sa_wg2", 0)
        return out

    old, new, again = run(4), run(0), run(0)
    assert float(old["dxkv"].float().abs().max()) > 0 and float(old["da1"].float().abs().max()) > 0
    for k in ("dxkv", "da1", "pg"):
        assert torch.equal(old[k], new[k]), (k, rel(new[k], old[k]))
        assert torch.equal(new[k], again[k]), k
    ops.clear_managed_shadows()


@pytest.mark.parametrize("B,G", [(128, 96), (3, 50)])
def test_ca_front_kernel_vs_the_separate_kernels(B, G):
    """vpf_ca_front_fwd (position MLP + tokens + pos + q_norm + q projection in one row-block kernel) against the kernels it replaces
    (vpf_smallk_fwd, vpf_gemm_h16 with bias, vpf_layernorm_fwd with the positional term, vpf_gemm_h16): same operands, same
    rounding points (h16 hidden layer, f32 pos / base, h16 q_norm output and q), agreement to rounding."""
    import torch.nn as nn
    from vipformer_amd import ops
    from vipformer_amd.model.pointcloud.partseg import CrossAttentionLayer
    D = 256
    torch.manual_seed(11)
    ca = CrossAttentionLayer(4, D, D, D, 2, 0.0, 0.1, 0.5).cuda()
    seq = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, D)).cuda()
    with torch.no_grad():
        ca[0].module.q_norm.weight.copy_(rnd(1, D) * 0.2 + 1.0); ca[0].module.q_norm.bias.copy_(rnd(2, D) * 0.1)
    centers = rnd(3, B, G, 3)
    tokens = rnd(4, B, G, D)

    class Enc:                      # what CaFrontFn / ca_front_supported look at
        num_cross_attention_layers = 1
        cross_attn_1 = ca
        sa_layers = []
    assert ops.ca_front_supported(seq, tokens, Enc)
    pos = ops.CaFrontFn.apply(centers, seq, tokens, Enc, *seq.parameters())
    st = ca.__dict__.pop("_vpf_front_stash")
    torch.cuda.synchronize()
    pos_ref = ops.PosMLPFn.apply(centers, seq, *seq.parameters())
    lnq = ca[0].module.q_norm
    nq, mq, rq, base = ops.layernorm_fwd(tokens.contiguous(), lnq.weight.data, lnq.bias.data, pos=pos_ref, want_sum=True)
    catt = ca[0].module.attention
    w16 = ops.shadow([catt.q_proj.weight, catt.k_proj.weight, catt.v_proj.weight])
    q = ops.linear_fwd(nq, w16[:D * D], D, D)
    torch.cuda.synchronize()
    # (the hidden layer's GELU inputs differ in the last bit -- the compiler contracts the three taps differently -- which flips a
    #  h16 rounding of the hidden activation here and there)
    assert rel(pos, pos_ref) < 5e-4, rel(pos, pos_ref)
    assert rel(st["base"].view(B, G, D), base.view(B, G, D)) < 5e-4
    assert rel(st["mq"], mq) < 1e-5 and rel(st["rq"], rq) < 1e-5
    assert rel(st["nq"].float(), nq.float()) < 2e-3, rel(st["nq"].float(), nq.float())
    assert rel(st["q"].float(), q.float()) < 4e-3, rel(st["q"].float(), q.float())
    # backward of the position MLP is PosMLPFn's own
    seq.zero_grad()
    R = rnd(5, B, G, D)
    (pos * R).sum().backward()
    g1 = [p.grad.clone() for p in seq.parameters()]
    seq.zero_grad()
    (pos_ref * R).sum().backward()
    for a_, b_ in zip(g1, [p.grad for p in seq.parameters()]):
        assert rel(a_, b_) < 1e-3          # (the same kernels on hidden activations that differ in a few h16 roundings; fp32 atomics)
