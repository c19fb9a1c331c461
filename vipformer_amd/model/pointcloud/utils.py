"""MI355X mirror of the reference's ``vipformer.model.pointcloud.utils`` (same names,
argument meaning and return contracts); every function dispatches to the C ABI of
libvipformer_hip.so.  Reference line numbers: vipformer/model/pointcloud/utils.py.

Differences that are deliberate and documented in DESIGN.md:
  * knn_point / divide_patches return neighbours in CANONICAL order (ascending
    distance, ties -> lower index); the reference's torch.topk(sorted=False) order
    is unspecified (and differs between torch CPU and GPU).
  * fp32 only, on an MI355X only (no CPU fallback).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import _lib as L


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def farthest_point_sample(pts: torch.Tensor, npoint: int) -> torch.Tensor:
    """utils.py:56-85.  pts [B,N,C] -> int64 [B,npoint].  Consumes torch's RNG exactly
    like the reference: one ``torch.randint(0, N, (B,))`` on pts.device (:71)."""
    L.need_cuda(pts)
    B, N, C = pts.shape
    start = torch.randint(0, N, (B,), dtype=torch.long, device=pts.device)
    return _fps_from_start(pts, npoint, start)


def _fps_from_start(pts: torch.Tensor, npoint: int, start: torch.Tensor) -> torch.Tensor:
    L.need_cuda(pts, start)
    pts = _f32c(pts.detach())
    B, N, C = pts.shape
    out = torch.empty(B, npoint, dtype=torch.long, device=pts.device)
    L.check(L.lib().vpf_fps_f32(L.ptr(pts), B, N, C, L.ptr(start.contiguous()), npoint, L.ptr(out), L.stream_ptr()),
            "vpf_fps_f32")
    return out


def index_points(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """utils.py:88-104.  points [B,N,C]; idx int64 [B,S] or [B,S,K] -> [B,S,C] / [B,S,K,C]."""
    L.need_cuda(points, idx)
    points = _f32c(points.detach())
    B, N, C = points.shape
    flat = idx.reshape(B, -1).contiguous()
    S = flat.shape[1]
    out = torch.empty(B, S, C, dtype=torch.float32, device=points.device)
    L.check(L.lib().vpf_index_points_f32(L.ptr(points), B, N, C, L.ptr(flat), S, L.ptr(out), L.stream_ptr()),
            "vpf_index_points_f32")
    return out.reshape(*idx.shape, C)


def fps(pts: torch.Tensor, number: int) -> torch.Tensor:
    """utils.py:41-53."""
    return index_points(pts, farthest_point_sample(pts, number))


def square_distance(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """utils.py:122-141 (3-channel recipe; extra channels are ignored like the callers do)."""
    L.need_cuda(src, dst)
    src, dst = _f32c(src.detach()), _f32c(dst.detach())
    B, Ns, Cs = src.shape
    _, Nd, Cd = dst.shape
    out = torch.empty(B, Ns, Nd, dtype=torch.float32, device=src.device)
    L.check(L.lib().vpf_square_distance_f32(L.ptr(src), Cs, L.ptr(dst), Cd, B, Ns, Nd, L.ptr(out), L.stream_ptr()),
            "vpf_square_distance_f32")
    return out


def _knn_group(xyz, centers, K, quirk, want_idx, want_dist, want_nb):
    L.need_cuda(xyz, centers)
    xyz, centers = _f32c(xyz.detach()), _f32c(centers.detach())
    B, N, C = xyz.shape
    _, G, Cc = centers.shape
    dev = xyz.device
    idx = torch.empty(B, G, K, dtype=torch.long, device=dev) if want_idx else None
    dist = torch.empty(B, G, K, dtype=torch.float32, device=dev) if want_dist else None
    nb = torch.empty(B, G, K, C, dtype=torch.float32, device=dev) if want_nb else None
    L.check(L.lib().vpf_knn_group_f32(L.ptr(xyz), B, N, C, L.ptr(centers), Cc, G, K, int(quirk), L.ptr(idx),
                                      L.ptr(dist), L.ptr(nb), L.stream_ptr()), "vpf_knn_group_f32")
    return idx, dist, nb


def knn_point(nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
    """utils.py:107-119 -> int64 [B,S,nsample], canonical order."""
    return _knn_group(xyz, new_xyz, nsample, False, True, False, False)[0]


def divide_patches(points: torch.Tensor, num_groups: int, group_size: int):
    """utils.py:6-38 -> (neighbors [B,G,K,C], centers [B,G,C]).  Reproduces the :36
    member-axis centre subtraction (members 0..2 only, all channels)."""
    centers = fps(points, num_groups)
    _, _, nb = _knn_group(points, centers, group_size, True, False, False, True)
    return nb, centers


# ----------------------------------------------------------------------------------------- modules
class Group2Emb(nn.Module):
    """utils.py:144-189: per-group mini-PointNet  [B,G,K,C] -> [B,G,dim_model].

    Same sub-module names as the reference (``first_conv.{0,1,3}``, ``second_conv.{0,1,3}``;
    conv weights are [out,in,1]) so checkpoints interchange; forward is one fused HIP
    sequence (K=3 front in VALU, 64->128 / 256->256 / 256->D on MFMA, BatchNorm batch
    statistics in fp32, both max-pools)."""

    def __init__(self, dim_model, point_channels=3):
        super().__init__()
        self.dim_model = dim_model
        self.point_channels = point_channels
        self.first_conv = nn.Sequential(nn.Conv1d(point_channels, 64, 1), nn.BatchNorm1d(64), nn.ReLU(inplace=True),
                                        nn.Conv1d(64, 128, 1))
        self.second_conv = nn.Sequential(nn.Conv1d(256, 256, 1), nn.BatchNorm1d(256), nn.ReLU(inplace=True),
                                         nn.Conv1d(256, self.dim_model, 1))

    def forward(self, point_groups):
        from ... import ops
        if point_groups.shape[-1] != self.point_channels:
            raise ValueError(f"expected {self.point_channels} point channels, got {point_groups.shape[-1]}")
        if self.dim_model % 8:
            raise L.VpfError("dim_model must be a multiple of 8")
        return ops.Group2EmbFn.apply(point_groups, self, self.training, *self.parameters())


class Sequential(nn.Sequential):
    """utils.py:245-252: nn.Sequential whose forward splats tuples into the next module."""

    def forward(self, *x):
        for layer in self:
            x = layer(*x) if type(x) == tuple else layer(x)
        return x


class PointNetFeaturePropagation(nn.Module):
    """utils.py:192-242 with the reference's layouts: xyz1 [B,C,N], xyz2 [B,C,S], points1 [B,D1,N] or None, points2 [B,D2,S]
    -> [B,mlp[-1],N].  3-NN inverse-distance interpolation (a running 3-minimum instead of the full sort of :224) + concat +
    (conv1d(k=1) + BatchNorm + ReLU) per entry of ``mlp``, through ops_seg.FeaturePropFn."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv1d(last_channel, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out_channel))
            last_channel = out_channel

    def forward(self, xyz1, xyz2, points1, points2):
        from ... import ops_seg as S
        x1, x2 = xyz1.permute(0, 2, 1), xyz2.permute(0, 2, 1)
        p1 = points1.permute(0, 2, 1) if points1 is not None else None
        y = S.FeaturePropFn.apply(x1, x2, p1, points2.permute(0, 2, 1), self, self.training, *self.parameters())
        return y.float().permute(0, 2, 1)
