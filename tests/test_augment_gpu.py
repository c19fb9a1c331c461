"""On-device augmentation against the oracle's restatement of the reference pipeline (oracle/augment.py, pinned
bit-exactly against the reference's data_utils.py): the deterministic part replayed from the exported per-cloud draws to fp32
rounding, the random parts (jitter, input dropout, flips) statistically; and the uint8 image path exactly."""
import numpy as np
import pytest
import torch

from tests import helpers as Hh

pytestmark = pytest.mark.gpu


def test_augment_points_replays_the_reference_pipeline():
    from oracle import augment as A
    from vipformer_amd import augment as G
    B, N = 64, 1024
    rng = np.random.default_rng(3)
    raw = (rng.standard_normal((B, N, 3)) * np.array([1.0, 0.4, 2.0]) + 0.7).astype(np.float32)
    out, params = G.augment_points(torch.from_numpy(raw).cuda(), return_params=True)
    out, params = out.cpu().numpy(), params.cpu().numpy().astype(np.float64)
    out2 = G.augment_points(torch.from_numpy(raw).cuda()).cpu().numpy()
    assert not np.array_equal(out, out2)                                    # a fresh draw per call (the two views of a pair)
    scale, angle, t, ratio = params[:, 0], params[:, 1], params[:, 2:5], params[:, 5]
    assert 0.5 <= scale.min() and scale.max() <= 2.0 and 0 <= angle.min() and angle.max() < 2 * np.pi + 1e-6
    assert np.abs(t).max() <= 0.5 and 0 <= ratio.min() and ratio.max() <= 0.875
    # the draws are spread over their ranges (64 clouds): means within 4 sigma of uniform
    assert abs(scale.mean() - 1.25) < 4 * (1.5 / np.sqrt(12 * B)) and abs(ratio.mean() - 0.4375) < 4 * (0.875 / np.sqrt(12 * B))
    jit_all, frac_dropped = [], []
    for b in range(B):
        # deterministic part with the SAME draws, in the oracle's order of operations (Normalize, Scale, Rotate, Translate)
        xyz = raw[b].astype(np.float32)
        xyz = xyz - np.mean(xyz, axis=0)
        xyz = xyz / np.max(np.sqrt(np.sum(xyz ** 2, axis=1)))
        xyz = xyz * np.float32(scale[b])
        R = A.angle_axis(float(angle[b]), np.array([0.0, 1.0, 0.0])).numpy()
        xyz = xyz @ R.T
        diff = xyz.max(0) - xyz.min(0)
        xyz = xyz + (t[b] * diff).astype(np.float32)
        d = out[b] - xyz
        dropped = np.all(out[b] == out[b][0], axis=1)
        dropped[0] = False
        keep = ~dropped
        keep[0] = True
        jit = d[keep]
        assert np.abs(jit).max() <= 0.05 + 2e-5, (b, np.abs(jit).max())      # |jitter| <= clip (+ fp32 rounding of the affine part)
        jit_all.append(jit)
        frac_dropped.append(dropped.mean())
        assert abs(dropped.mean() - ratio[b]) < 4 * np.sqrt(max(ratio[b] * (1 - ratio[b]), 1e-4) / N) + 2.0 / N, (b, dropped.mean(), ratio[b])
    jit = np.concatenate(jit_all)
    assert abs(jit.mean()) < 2e-4 and abs(jit.std() - 0.01) < 3e-4, (jit.mean(), jit.std())     # N(0, 0.01), clipping at 5 sigma is invisible
    # the same moments from the oracle's (reference-pinned) trans_1 on one cloud: same distribution family
    np.random.seed(0); torch.manual_seed(0)
    ref = A.trans_1(raw[0].copy()).numpy()
    assert ref.shape == (N, 3) and np.isfinite(ref).all()


def test_image_u8_normalize_exact():
    from vipformer_amd import augment as G
    rng = np.random.default_rng(5)
    img = (rng.random((16, 224, 224, 3)) * 256).astype(np.uint8)
    out, flips = G.image_u8_normalize(torch.from_numpy(img).cuda(), p_flip=0.5, return_flips=True)
    out, flips = out.cpu().numpy(), flips.cpu().numpy().astype(bool)
    assert 2 <= flips.sum() <= 14                                           # 16 draws at p = 0.5
    mean, std = np.array(G.IMAGENET_MEAN, np.float32), np.array(G.IMAGENET_STD, np.float32)
    x = img.astype(np.float32) / 255.0
    x[flips] = x[flips][:, :, ::-1]
    ref = ((x - mean) / std).transpose(0, 3, 1, 2)
    assert np.abs(out - ref).max() < 2e-6
    noflip = G.image_u8_normalize(torch.from_numpy(img).cuda(), p_flip=0.0).cpu().numpy()
    assert np.abs(noflip - (((img.astype(np.float32) / 255.0) - mean) / std).transpose(0, 3, 1, 2)).max() < 2e-6


def test_augmented_batch_feeds_the_training_step():
    """Raw clouds + uint8 images -> on-device augmentation -> Pretrainer.step: the pipeline of pretrain.py:173-211 with the
    DataLoader-worker work moved to the GPU."""
    from vipformer_amd import augment as G
    from vipformer_amd.train import Pretrainer, build_models
    a = Hh.ARCHS["tiny"]
    pc, im = build_models(D=a["D"], H=a["H"], G=a["G"], K=a["K"], S=a["S"], MR=a["MR"], N=a["N"], img=a["img"], patch=a["patch"])
    pc.train(); im.train()
    tr = Pretrainer(pc, im)
    raw = Hh.synth_points(1, 8, a["N"]).cuda() * 3.0 + 1.0
    img = (torch.rand(8, a["img"], a["img"], 3) * 255).to(torch.uint8).cuda()
    t1, t2 = G.augment_points(raw), G.augment_points(raw)
    imgs = G.image_u8_normalize(img)
    loss, _, _ = tr.step(t1, t2, imgs)
    torch.cuda.synchronize()
    assert torch.isfinite(loss).item() and torch.isfinite(tr.flat.g).all().item()
