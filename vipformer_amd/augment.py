"""On-device augmentation: what the reference's DataLoader workers do per sample on the CPU
(datasets/data.py:16-36,97-112) as two HIP kernels, so the host ships raw clouds and uint8 images.

    t1 = augment_points(raw)            # trans_1: one draw per call -> call twice for the two views (data.py:107-108)
    t2 = augment_points(raw)
    imgs = image_u8_normalize(img_u8)   # ToTensor + Normalize(ImageNet) + RandomHorizontalFlip -> f32 [B,3,H,W] (pretrain.py:177-179)

The random draws come from the library's counter-based stream (ops.rng): statistically, not bitwise, the reference's numpy / torch
draws -- the deterministic part is replayable from the exported per-cloud parameters (tests/test_augment_gpu.py)."""
from __future__ import annotations

import ctypes

import torch

from . import _lib as L
from . import ops

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # utils.py:25
IMAGENET_STD = (0.229, 0.224, 0.225)
_SITE_POINTS, _SITE_FLIP = 0xA0610001, 0xA0610002


def augment_points(pts: torch.Tensor, return_params: bool = False, state: torch.Tensor = None):
    """pts f32 [B,N,C>=3] (raw clouds on the GPU) -> f32 [B,N,3] after trans_1 (datasets/data.py:16-25).  Every call draws a fresh
    set of parameters (``ops.rng.acquire``) unless ``state`` (a 4 x int32 rng state) is given."""
    L.need_cuda(pts)
    B, N, C = pts.shape
    x = pts.detach().contiguous().float()
    st = state if state is not None else ops.rng.acquire(x.device, True)
    out = torch.empty(B, N, 3, dtype=torch.float32, device=x.device)
    params = torch.empty(B, 8, dtype=torch.float32, device=x.device) if return_params else None
    L.call("vpf_augment_points", x, B, N, C, st, _SITE_POINTS, out, params)
    return (out, params) if return_params else out


def image_u8_normalize(img_u8: torch.Tensor, mean=IMAGENET_MEAN, std=IMAGENET_STD, p_flip: float = 0.5, return_flips: bool = False,
                       state: torch.Tensor = None):
    """img_u8 uint8 [B,H,W,3] on the GPU -> f32 [B,3,H,W]: /255, Normalize(mean, std), RandomHorizontalFlip(p_flip) (utils.py:21-25;
    Resize and ColorJitter stay with the decoder on the host: they work on PIL images)."""
    L.need_cuda(img_u8)
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 4 or img_u8.shape[-1] != 3:
        raise L.VpfError("image_u8_normalize expects uint8 [B,H,W,3]")
    B, H, W, _ = img_u8.shape
    x = img_u8.contiguous()
    st = (state if state is not None else ops.rng.acquire(x.device, True)) if p_flip > 0 else None
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=x.device)
    flips = torch.empty(B, dtype=torch.uint8, device=x.device) if return_flips else None
    m, s = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    L.call("vpf_image_u8_normalize", x, B, H, W, m, s, st, _SITE_FLIP, float(p_flip), out, flips)
    return (out, flips) if return_flips else out
