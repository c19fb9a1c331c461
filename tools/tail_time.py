#!/usr/bin/env python3
"""time the dominant kernel (fused encoder-layer tail) alone, exactly as bench.py's roofline leg does: tools/tail_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
r = bench.time_dominant_kernel(torch.device("cuda:0"))
print({k: r[k] for k in ("us_per_launch", "achieved", "frac")})
