"""Shared, machine-portable synthetic data for tests, golden generation and bench.

Everything is derived from numpy's PCG64 ``Generator.random()`` (53-bit integer ->
double, exact) so the same seed gives the same bits in the build container (where the
reference is importable and the golden fixtures are made) and on the GPU box (where
it is not).  No torch RNG, no libm.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, List, Tuple

import numpy as np
import torch

TEST_LOSS_SCALE = 256.0       # = tests/conftest.py's VPF_LOSS_SCALE: GradScaler's scale after it has backed off on the tests' tiny batches
H16 = torch.float16          # the library's 16-bit operand dtype (vipformer_amd._lib.H16; a GPU test checks they agree)

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _uniform(rng: np.random.Generator, shape, lo: float, hi: float) -> torch.Tensor:
    a = rng.random(tuple(shape)) * (hi - lo) + lo
    return torch.from_numpy(a.astype(np.float32))


def synth_state_dict(keyshapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int, alias_ca: bool = True) -> Dict[str, torch.Tensor]:
    """Deterministic, well-conditioned values for a reference-shaped state dict.
    Matrices ~U(+-sqrt(3/fan_in)) (unit gain), norm scales 1+-0.2, shifts/biases +-0.1,
    running_var in [0.5,1.5].  ``encoder.cross_attn_1.*`` aliases ``cross_attn_n.*``
    (one parameter set in the reference, partseg.py:297-298) unless alias_ca is False (num_cross_attention_layers > 1)."""
    rng = np.random.default_rng(seed)
    out: Dict[str, torch.Tensor] = {}
    ks = sorted(keyshapes, key=lambda t: t[0])
    for key, shape in ks:
        shape = tuple(shape)
        if alias_ca and "cross_attn_1." in key:
            continue
        if key.endswith("num_batches_tracked"):
            out[key] = torch.zeros((), dtype=torch.int64)
        elif key.endswith("running_var"):
            out[key] = _uniform(rng, shape, 0.5, 1.5)
        elif key.endswith("running_mean"):
            out[key] = _uniform(rng, shape, -0.1, 0.1)
        elif key == "position_emb":
            out[key] = _uniform(rng, shape, -1.0, 1.0)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            b = float(np.sqrt(3.0 / fan_in))
            out[key] = _uniform(rng, shape, -b, b)
        elif key.endswith("weight"):           # LayerNorm / BatchNorm scale
            out[key] = _uniform(rng, shape, 0.8, 1.2)
        else:                                  # biases, norm shifts
            out[key] = _uniform(rng, shape, -0.1, 0.1)
    for key, shape in ks:
        if alias_ca and "cross_attn_1." in key:
            out[key] = out[key.replace("cross_attn_1.", "cross_attn_n.")]
    return out


def synth_points(seed: int, B: int, N: int, C: int = 3, mode: str = "uniform") -> torch.Tensor:
    """uniform: U(-0.57,0.57)^C (inside the unit sphere, like PointcloudNormalize output);
    dups: a random 0..87.5% of rows replaced by row 0 (RandomInputDropout, data_utils.py:185-188);
    grid: coordinates k*2^-8, |k|<=256 (every product/sum exact in fp32)."""
    rng = np.random.default_rng(seed)
    if mode == "grid":
        k = np.floor(rng.random((B, N, C)) * 513.0) - 256.0
        return torch.from_numpy((k / 256.0).astype(np.float32))
    pts = (rng.random((B, N, C)) * 2.0 - 1.0) * 0.57
    pts = pts.astype(np.float32)
    if mode == "dups":
        for b in range(B):
            ratio = rng.random() * 0.875
            drop = np.nonzero(rng.random(N) <= ratio)[0]
            if drop.size:
                pts[b, drop, :] = pts[b, 0, :]
    return torch.from_numpy(pts)


def synth_start(seed: int, B: int, N: int) -> torch.Tensor:
    rng = np.random.default_rng(seed + 7919)
    return torch.from_numpy(np.floor(rng.random(B) * N).astype(np.int64))


def synth_images(seed: int, B: int, H: int, W: int) -> torch.Tensor:
    """[B,H,W,3] view of an NCHW tensor exactly like pretrain.py:179 hands it over;
    ~zero-mean/unit-variance like post-Normalize images."""
    rng = np.random.default_rng(seed + 104729)
    a = (rng.random((B, 3, H, W)) * 2.0 - 1.0) * np.sqrt(3.0)
    return torch.from_numpy(a.astype(np.float32)).permute(0, 2, 3, 1)


def synth_like(seed: int, shape) -> torch.Tensor:
    rng = np.random.default_rng(seed + 15485863)
    return _uniform(rng, shape, -1.0, 1.0)


def padmask_inputs():
    """Inputs of the pad_mask fixture padmask.npz (tests/golden/make_golden.py make_padmask).  Ragged lengths (40 queries, 70 keys:
    neither a multiple of 32); batch row 0 pads its last 20 keys, row 1 a scattered half, row 2 EVERY key (softmax of equal scores:
    uniform attention, no gradient to q / k).  pad_self: the mask of a self-attention over the 40 query tokens."""
    B, Lq, Lk, D, H = 3, 40, 70, 128, 2
    rng = np.random.default_rng(4242)
    pad = np.zeros((B, Lk), dtype=bool)
    pad[0, 50:] = True
    pad[1] = rng.random(Lk) < 0.5
    pad[2] = True
    pad_self = np.zeros((B, Lq), dtype=bool)
    pad_self[0, 33:] = True
    pad_self[1] = rng.random(Lq) < 0.4
    pad_self[2] = True
    return dict(B=B, Lq=Lq, Lk=Lk, D=D, H=H, xq=synth_like(900, (B, Lq, D)), xkv=synth_like(901, (B, Lk, D)),
                pos=0.5 * synth_like(902, (B, Lq, D)), R=synth_like(903, (B, Lq, D)),
                pad=torch.from_numpy(pad), pad_self=torch.from_numpy(pad_self))


def load_keyshapes(name: str) -> List[Tuple[str, Tuple[int, ...]]]:
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        return [(k, tuple(s)) for k, s in json.load(f)]


def golden(name: str):
    return np.load(os.path.join(GOLDEN_DIR, name))


# tiny architectures shared by the golden generator and the tests
ARCHS = {
    # name: (D, H, G, K, S, MR, N, img, patch)
    "tiny": dict(D=64, H=1, G=16, K=8, S=2, MR=2, N=256, img=32, patch=8),
    "tiny2": dict(D=128, H=2, G=32, K=16, S=1, MR=4, N=512, img=64, patch=16),
    "c1": dict(D=256, H=4, G=96, K=32, S=6, MR=2, N=1024, img=224, patch=16),
    "c3": dict(D=256, H=4, G=128, K=32, S=8, MR=2, N=1024, img=224, patch=16),
    "c4": dict(D=384, H=6, G=128, K=32, S=8, MR=4, N=2048, img=224, patch=16),
    "tinyseg": dict(D=64, H=1, G=16, K=8, S=3, MR=2, N=256, img=32, patch=8),       # CrossFormer_partseg needs >= 3 layers to tap
    # the geometry the reference's own training scripts ship (scripts/pretrain/pt-E1CL6SL-H4D256-L96-MR2-0.sh:10-16 with parser.py:112's
    # default patch_size 12): 2048-point clouds at D = 256, 144 x 144 images -> T = 144 tokens, patch row K = 432 (not a multiple of 64)
    "ref144": dict(D=256, H=4, G=96, K=32, S=6, MR=2, N=2048, img=144, patch=12),
    "ref144m4": dict(D=256, H=4, G=96, K=32, S=6, MR=4, N=2048, img=144, patch=12),  # ...-MR4-0.sh: hidden 1024 at D = 256
}
REF_ARCHS = ("ref144", "ref144m4")
FULLSIZE = ("c1", "c3", "c4") + REF_ARCHS          # fixtures of these hold slices

# pairs per batch used by the model_* / dropout-step fixtures and tests
MODEL_BATCH = {"tiny": 8, "tiny2": 8, "c1": 4, "c3": 4, "c4": 4, "tinyseg": 8, "ref144": 4, "ref144m4": 4}

# pairs PER GPU of BASELINE.json configs[1], [2], [3] (64 on one GPU; 256 and 128 over 8): the batch of the fullsize_*.npz fixtures
FULL_BATCH = {"c1": 64, "c3": 32, "c4": 16}
GRAD_SAMPLE = 512


def grad_sample(g: torch.Tensor, cap: int = GRAD_SAMPLE) -> torch.Tensor:
    """A strided sample of <= cap elements of a gradient tensor (flattened, every (n // cap)-th element from 0): what the fullsize_*
    fixtures keep of the reference's gradients and what the tests take from the HIP gradients."""
    flat = g.detach().reshape(-1)
    n = flat.numel()
    return flat[::max(1, n // cap)][:cap].float().cpu().clone()


def partseg_inputs(name: str, B: int, seed: int):
    """Inputs of a part-segmentation fixture: clouds, FPS start indices, one-hot object labels (16 classes), per-point part targets (50)."""
    a = ARCHS[name]
    pts = synth_points(seed, B, a["N"]); start = synth_start(seed, B, a["N"])
    cls = torch.zeros(B, 16)
    cls[torch.arange(B), torch.arange(B) % 16] = 1.0
    target = torch.from_numpy((np.random.default_rng(seed + 1).random((B, a["N"])) * 50).astype(np.int64))
    return pts, start, cls, target


# CrossFormer_partseg taps (1-based self-attention layer numbers; the reference needs 3 or 4 of them, partseg.py:430-435)
PARTSEG_LAYERS = {"tinyseg": [1, 2, 3], "c3": [2, 5, 8]}
