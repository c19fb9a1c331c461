"""The per-epoch SVM probe of pretrain.py:226-276 on the HIP path: eval-mode backbone features at probe batch
sizes equal the oracle's, are what ``pc_model(data)[1]`` returns batch by batch, and feed sklearn's linear SVC."""
import numpy as np
import pytest
import torch

from tests import helpers as Hh
from tests.test_modules_gpu import build, forced_start, rel

pytestmark = pytest.mark.gpu


def test_probe_features_match_oracle_and_feed_the_linear_svm():
    from oracle import torch_oracle as O
    from vipformer_amd import probe
    pc, im, a = build("tiny")
    N = a["N"]
    rng = np.random.default_rng(0)

    def clouds(n, kind):             # two separable "classes": points on a sphere shell vs in a flat slab
        p = rng.standard_normal((n, N, 3)).astype(np.float32)
        if kind == 0:
            p /= np.linalg.norm(p, axis=2, keepdims=True)
        else:
            p[:, :, 1] *= 0.05
            p /= np.abs(p).max(axis=(1, 2), keepdims=True)
        return torch.from_numpy(p * 0.57)

    def loader(n_per_class, bs):
        data = torch.cat([clouds(n_per_class, 0), clouds(n_per_class, 1)])
        label = torch.cat([torch.zeros(n_per_class, 1), torch.ones(n_per_class, 1)]).long()       # ModelNet40 style [B,1] labels
        perm = torch.from_numpy(rng.permutation(2 * n_per_class))
        data, label = data[perm], label[perm]
        return [(data[i:i + bs], label[i:i + bs]) for i in range(0, 2 * n_per_class, bs)], data, label

    train_batches, train_data, _ = loader(24, 16)
    test_batches, _, _ = loader(12, 8)
    pc.train()                                                     # the probe must switch to eval mode itself and restore
    starts = [Hh.synth_start(50 + i, len(b[0]), N).cuda() for i, b in enumerate(train_batches)]
    feats = []
    real = torch.randint
    it = iter(starts)
    torch.randint = lambda *x, **k: next(it).clone()
    try:
        f_train, y_train = probe.extract_features(pc, train_batches)
    finally:
        torch.randint = real
    assert pc.training and f_train.shape == (48, 2 * a["D"]) and y_train.shape == (48,)
    # batch 0 against the oracle (eval mode: running statistics, no dropout)
    sd = Hh.synth_state_dict(Hh.load_keyshapes("keys_pc_tiny.json"), 100)
    arch = O.Arch(D=a["D"], H=a["H"], G=a["G"], K=a["K"], S=a["S"], MR=a["MR"], img=a["img"], patch=a["patch"])
    with torch.no_grad():
        _, bb = O.pc_forward(sd, train_batches[0][0], starts[0].cpu(), arch, False)
    assert rel(torch.from_numpy(f_train[:16]), bb) < 2e-2
    f_test, y_test = probe.extract_features(pc, test_batches)
    acc = probe.svm_probe(f_train, y_train, f_test, y_test, C=1.0)
    assert acc >= 0.9, acc                                         # separable classes: even random-weight features split them
