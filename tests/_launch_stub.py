"""A stand-in for bench.py in tests/test_host_cpu.py::test_launch_ranks_*: N ranks over gloo on CPU tensors; rank 0 prints a JSON line,
then EVERY rank prints more text to stdout (what RCCL's banner does at exit).  argv: [mode]  mode = ok | fail | hang"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

mode = sys.argv[1] if len(sys.argv) > 1 else "ok"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.ones(1)
dist.all_reduce(t)
if mode == "fail" and rank == world - 1:
    sys.exit(7)
if mode == "hang" and rank == world - 1:
    time.sleep(3600)
if rank == 0:
    print(json.dumps({"metric": "stub", "ranks_seen": int(t.item()), "n_gpus": world}), flush=True)
dist.barrier()
print(f"banner of rank {rank}: text behind the result line", flush=True)
dist.destroy_process_group()
