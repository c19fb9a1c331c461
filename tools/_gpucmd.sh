#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "grouped_wgrad" --tb=line 2>&1 | grep -v "^E   *+\|tensor" | cut -c1-400 | tail -5
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, torch
sys.path.insert(0, "tests")
import conftest  # noqa
from test_boundary_gpu import build, _batch
from vipformer_amd.train import Pretrainer
for pairs in (2, 4, 8):
    pc, im, a = build("tiny", (0.1, 0.5)); pc.train(); im.train()
    tr = Pretrainer(pc, im, loss_scale=65536.0, growth_interval=1000)
    t1, t2, imgs, start = _batch(a, pairs)
    tr.capture(t1, t2, imgs.permute(0, 3, 1, 2).contiguous(), warmup=2)
    ls = [float(tr.replay()[0]) for _ in range(40)]
    print(pairs, "pairs: skipped", tr.skipped_steps, "scale", tr.loss_scale, "loss", ls[0], "->", ls[-1])
PY
