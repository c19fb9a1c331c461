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


# Collection order on the GPU box: the oracle / golden parity tests first, the hot path's stages before its
# compositions, full-size configurations next, plumbing after that and everything that starts processes last -- `pytest -x` then stops
# at the most informative failure, and multi-process plumbing can never shadow the parity suite.
_ORDER = ("test_oracle_golden", "test_host_cpu", "test_preproc_gpu", "test_kernels_gpu", "test_modules_gpu", "test_fullsize_gpu",
          "test_partseg_gpu", "test_probe_gpu", "test_trajectory_gpu", "test_augment_gpu", "test_boundary_gpu", "test_zz_multiproc_gpu")


def _rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    return _ORDER.index(name) if name in _ORDER else len(_ORDER) - 1


def pytest_collection_modifyitems(config, items):
    items.sort(key=_rank)                                    # stable: the order inside a file is kept
    import torch
    if torch.cuda.is_available():
        # The checker (oracle/torch_oracle.py) runs on the GPU box's HOST: with torch's default of one thread per core of a 100+-core
        # box its small fp32 ops are oversubscribed and run ~9 x slower than on 8 threads (bench.py's cpu_baseline measured exactly that:
        # 1.43 vs 12.7 pairs/s) -- round 5's suite spent 500 of its 535 s there.  8 threads, the container's count.
        torch.set_num_threads(min(8, os.cpu_count() or 8))
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _history_independent_random_state(request):
    """Every GPU test starts from ITS OWN fixed random state -- torch's generators (FPS start indices, torch.randn inputs) and the
    library's device-resident dropout state -- derived from the test's node id, so that a test's masks and inputs do not depend on which
    tests ran before it (round 6: re-ordering the suite moved `test_pretrainer_at_the_default_loss_scale_backs_off_and_trains` onto
    a marginal realisation of its 2-pair dropout masks, and it failed once in three suites).  Tests that seed explicitly are unaffected."""
    if "gpu" in request.keywords:
        import zlib
        import torch
        if torch.cuda.is_available():
            from vipformer_amd import ops
            h = zlib.crc32(request.node.nodeid.encode())
            torch.manual_seed(h)
            ops.rng.seed(0x5EED0000 + h)
    yield
