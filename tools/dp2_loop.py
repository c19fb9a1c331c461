#!/usr/bin/env python3
"""Soak test of tools/dp2_one_gpu.py: N fresh launches of the two-ranks-on-one-GPU rehearsal, each with a short
watchdog; every launch's per-rank logs (progress marks; on a stall the Python stacks of all threads and the native threads' wait states)
are kept under gpurun_out/dp2_loop/iter_XXX/, one summary line per launch in gpurun_out/dp2_loop/summary.txt.
usage: python tools/dp2_loop.py [launches=50] [pairs=4] [steps=2] [parent_gpu=1]
parent_gpu=1: this process initialises HIP and keeps 1 GB allocated first, as the pytest process that launches the tool does."""
import os, subprocess, sys, time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
pairs = sys.argv[2] if len(sys.argv) > 2 else "4"
steps = sys.argv[3] if len(sys.argv) > 3 else "2"
parent_gpu = (sys.argv[4] if len(sys.argv) > 4 else "1") == "1"
out = os.path.join(root, "gpurun_out", "dp2_loop")
os.makedirs(out, exist_ok=True)
if parent_gpu:
    import torch
    keep = torch.zeros(256 << 20, device="cuda")          # noqa: F841
    torch.cuda.synchronize()
env = dict(os.environ, DP2_WATCHDOG_S=os.environ.get("DP2_WATCHDOG_S", "60"))
bad = 0
with open(os.path.join(out, "summary.txt"), "a") as summ:
    for i in range(n):
        d = os.path.join(out, f"iter_{i:03d}")
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "dp2_one_gpu.py"), pairs, steps, d], capture_output=True, text=True, env=env)
        dt = time.time() - t0
        ok = r.returncode == 0 and "dp2 on one GPU: ok" in r.stdout
        bad += int(not ok)
        line = f"launch {i:3d}: {'ok  ' if ok else 'FAIL'} rc {r.returncode} {dt:6.1f} s"
        print(line, flush=True)
        summ.write(line + "\n"); summ.flush()
        if not ok:
            with open(os.path.join(d, "tool_stdout.txt"), "w") as f:
                f.write(r.stdout)
            with open(os.path.join(d, "tool_stderr.txt"), "w") as f:
                f.write(r.stderr[-20000:])
    line = f"{n - bad} of {n} launches ok"
    print(line); summ.write(line + "\n")
sys.exit(1 if bad else 0)
