#!/usr/bin/env python3
"""How much margin do the "the captured step trains" assertions have?  The scenarios of tests/test_boundary_gpu.py
(test_pretrainer_loss_scale_backs_off_on_overflow_inside_the_captured_graph: 8 pairs from 2 ** 30;
 test_pretrainer_at_the_default_loss_scale_backs_off_and_trains: 2 pairs from 65 536) under different dropout seeds:
loss at the initial weights (mean over the skipped replays), mean of the last 8 replays, the last one, skipped steps.
usage: python tools/diag_train_margin.py [seeds=12]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VPF_LOSS_SCALE", "256")
import torch
from tests import helpers as Hh
from tests.test_modules_gpu import build
from tests.test_boundary_gpu import _batch
from vipformer_amd import ops
from vipformer_amd.train import Pretrainer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for pairs, scale in ((8, 2.0 ** 30), (2, 65536.0)):
    for seed in range(n):
        ops.rng.seed(1000 + seed)
        pc, im, a = build("tiny", (0.1, 0.5))
        pc.train(); im.train()
        tr = Pretrainer(pc, im, loss_scale=scale, growth_interval=1000)
        t1, t2, imgs, start = _batch(a, pairs)
        tr.capture(t1, t2, imgs.permute(0, 3, 1, 2).contiguous(), warmup=2)
        losses = [float(tr.replay()[0]) for _ in range(40)]
        sk = tr.skipped_steps
        first = sum(losses[:max(sk, 1)]) / max(sk, 1)
        print(f"pairs {pairs} seed {seed}: skipped {sk:2d}  loss at the initial weights {first:.3f} (first replay {losses[0]:.3f})  last 8 mean {sum(losses[-8:]) / 8:.3f}  "
              f"last {losses[-1]:.3f}  min over last 8 {min(losses[-8:]):.3f} max {max(losses[-8:]):.3f}", flush=True)
