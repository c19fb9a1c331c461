import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# GradScaler's initial loss scale for every Pretrainer the tests build (train.Pretrainer reads VPF_LOSS_SCALE when none is passed).  The
# tests run 2 - 8 pairs and inspect ONE step: per-sample gradients are 8 - 30 x those of the benchmark's 64-pair batch, and torch's
# default 2 ** 16 overflows fp16 there -- which real training answers by skipping steps and halving the scale (tested on its own in
# test_kernels_gpu.py::test_adamw_follows_gradscaler_step_and_update and test_boundary_gpu.py); here the backed-off scale is set up front.
os.environ.setdefault("VPF_LOSS_SCALE", "256")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
