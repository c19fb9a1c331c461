"""GPU parity: FPS / kNN / grouping through the C ABI vs the C oracle and the golden
fixtures (bit-exact), plus size-independent properties at the BASELINE sizes."""
import numpy as np
import pytest
import torch

from tests import helpers as Hh
from tests.helpers import H16

pytestmark = pytest.mark.gpu
PRE = ["u1024", "u2048", "d1024", "g1024", "c6_1000", "u256", "u512"]


@pytest.mark.parametrize("tag", PRE)
def test_preproc_golden_bit_exact(tag):
    from vipformer_amd.model.pointcloud import utils as U
    g = Hh.golden(f"preproc_{tag}.npz")
    seed, B, N, C, G, K = [int(v) for v in g["meta"]]
    mode = {"d": "dups", "g": "grid"}.get(tag[0], "uniform")
    pts = Hh.synth_points(seed, B, N, C, mode).cuda()
    start = Hh.synth_start(seed, B, N).cuda()
    idx = U._fps_from_start(pts, G, start)
    assert np.array_equal(idx.cpu().numpy(), g["fps_idx"])
    centers = U.index_points(pts, idx)
    assert np.array_equal(centers.cpu().numpy(), g["centers"])
    d = U.square_distance(centers, pts).cpu().numpy().view(np.uint32)
    gb = g["sqdist_bits"]
    assert np.array_equal(d[:, :gb.shape[1], :], gb)
    kidx, kdist, nb = U._knn_group(pts, centers, K, True, True, True, True)
    assert np.array_equal(kidx.cpu().numpy(), g["knn_canonical"])
    assert np.array_equal(kdist.cpu().numpy().view(np.uint32), g["knn_dist"].view(np.uint32))
    assert np.array_equal(nb.cpu().numpy(), g["neighbors"])
    assert np.array_equal(U.knn_point(K, pts[:, :, :3], centers[:, :, :3]).cpu().numpy(), g["knn_canonical"])


@pytest.mark.parametrize("B,N,G,K,mode", [(128, 1024, 96, 32, "uniform"), (128, 1024, 96, 32, "dups"),
                                          (64, 1024, 128, 32, "uniform"), (32, 2048, 128, 32, "uniform"),
                                          (5, 777, 33, 17, "uniform"), (3, 64, 64, 64, "grid"), (2, 3000, 50, 32, "uniform")])
def test_preproc_vs_oracle_full_size(B, N, G, K, mode):
    """BASELINE sizes (c2: 128 clouds x 1024 pts, c3, c4) and ragged sizes vs the C oracle."""
    from oracle import torch_oracle as O
    from vipformer_amd.model.pointcloud import utils as U
    pts = Hh.synth_points(900 + B + N, B, N, 3, mode)
    start = Hh.synth_start(901, B, N)
    ref_idx = O.fps_indices(pts, start, G)
    nb_ref, ct_ref, kidx_ref = O.divide_patches(pts, ref_idx, K, True)
    dp = pts.cuda()
    idx = U._fps_from_start(dp, G, start.cuda())
    assert torch.equal(idx.cpu(), ref_idx)
    centers = U.index_points(dp, idx)
    kidx, _, nb = U._knn_group(dp, centers, K, True, True, False, True)
    assert torch.equal(kidx.cpu(), kidx_ref)
    assert torch.equal(nb.cpu(), nb_ref) and torch.equal(centers.cpu(), ct_ref)
    # properties: FPS indices distinct unless the cloud has duplicates; first kNN member is the centre itself
    if mode == "uniform":
        assert all(len(set(r.tolist())) == G for r in idx.cpu())
        assert torch.equal(kidx[:, :, 0].cpu(), idx.cpu())


def test_divide_patches_signature_and_rng():
    """divide_patches(points,G,K) draws exactly one randint like utils.py:71."""
    from vipformer_amd.model.pointcloud import utils as U
    pts = Hh.synth_points(5, 4, 512).cuda()
    torch.manual_seed(3)
    nb, ct = U.divide_patches(pts, 32, 16)
    after = torch.randint(0, 10, (1,), device="cuda").item()
    torch.manual_seed(3)
    start = torch.randint(0, 512, (4,), dtype=torch.long, device="cuda")
    assert torch.randint(0, 10, (1,), device="cuda").item() == after
    assert nb.shape == (4, 32, 16, 3) and ct.shape == (4, 32, 3)
    idx = U._fps_from_start(pts, 32, start)
    assert torch.equal(ct, U.index_points(pts, idx))


def test_errors_are_loud():
    from vipformer_amd import _lib
    from vipformer_amd.model.pointcloud import utils as U
    with pytest.raises(_lib.VpfError):
        U.knn_point(4, torch.zeros(1, 8, 3), torch.zeros(1, 2, 3))          # CPU tensors: no fallback
    with pytest.raises(_lib.VpfError):
        U.knn_point(65, torch.zeros(1, 128, 3).cuda(), torch.zeros(1, 2, 3).cuda())   # K > 64


def test_fps_is_stable_beside_gemms_on_another_stream():
    """fps_kernel sharing CUs with gemm_kernel workgroups (MFMAs fed by LDS fragment reads) used to return a wrong sampling in a third
    or more of the launches (the packed-fp32 instructions of its distance update; preproc.hip is compiled without
    them now).  The two-stream reproduction: the sampling must be the C oracle's, launch after launch."""
    from oracle import torch_oracle as O
    from vipformer_amd import ops
    from vipformer_amd.model.pointcloud import utils as U
    pts = Hh.synth_points(21, 128, 1024)
    start = Hh.synth_start(21, 128, 1024)
    want = O.fps_indices(pts, start, 96)
    pts_d, start_d = pts.cuda(), start.cuda()
    M, N, K = 12288, 512, 256
    dY = torch.randn(M, N, device="cuda").to(H16)
    W = (torch.randn(N, K, device="cuda") * 0.05).to(H16)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    bad = 0
    for _ in range(300):
        with torch.cuda.stream(side):
            for _ in range(12):
                ops.linear_dgrad(dY, W, N, K)
        idx = U._fps_from_start(pts_d, 96, start_d)
        torch.cuda.synchronize()
        bad += int(not torch.equal(idx.cpu(), want))
    assert bad == 0, f"{bad} of 300 launches beside dgrad GEMMs differ from the oracle's sampling"
