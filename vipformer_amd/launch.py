"""One process per GPU, started from a plain ``python bench.py --gpus N`` (the reference self-spawns too: pretrain.py:332-341,
``mp.spawn(main, nprocs=world_size)``).

``launch_ranks`` must be called BEFORE the calling process has touched the GPU (no HIP call, no ``torch.cuda.is_available()``): it
starts ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P script args...`` as a
fresh child in its own session, relays the ranks' output, and returns the exit code.  Contract of the relay: every line the ranks wrote to
stdout goes to this process's stderr as it arrives, except that the LAST line that parses as a JSON object is held back and printed to
stdout at the very end -- so a driver reading the last line of stdout finds rank 0's result line whatever RCCL, torchrun or another rank
printed after it.  A launch that does not finish within ``timeout_s`` is ended by killing the session this function started (its exact
process group, never a pattern) and reported with exit code 124; the ranks' own watchdogs (bench.py: faulthandler) have printed their
stacks to stderr by then.  Nothing is ever re-executed in place."""
from __future__ import annotations

import json
import os
import signal
import socket
import subprocess
import sys
import threading
from typing import List, Optional, Sequence


def _ephemeral_port() -> int:
    """A TCP port for torchrun's rendezvous store.  Bound to port 0 and released just before the child binds it: a window exists, and
    a collision ends the launch with torchrun's own error and a non-zero exit code (never a hang: the store's bind fails at once)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(script: str, script_args: Sequence[str], nproc: int, timeout_s: Optional[float] = None,
                 env: Optional[dict] = None, port: Optional[int] = None) -> int:
    cmd: List[str] = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(nproc)}",
                      "--master-addr", "127.0.0.1", "--master-port", str(port or _ephemeral_port()), script, *script_args]
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this host driver
    e.setdefault("OMP_NUM_THREADS", "1")                     # (torchrun sets it anyway and says so on stderr)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=e, start_new_session=True)
    held: List[Optional[str]] = [None]

    def pump():
        for line in proc.stdout:
            line = line.rstrip("\n")
            is_json = False
            if line.startswith("{") and line.endswith("}"):
                try:
                    is_json = isinstance(json.loads(line), dict)
                except ValueError:
                    is_json = False
            if is_json:
                if held[0] is not None:
                    print(held[0], file=sys.stderr, flush=True)
                held[0] = line
            else:
                print(line, file=sys.stderr, flush=True)

    t = threading.Thread(target=pump, daemon=True)
    t.start()
    try:
        rc = proc.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        print(f"[launch] {nproc} ranks did not finish within {timeout_s:.0f} s: ending the session this launcher started (pgid {proc.pid})",
              file=sys.stderr, flush=True)
        try:
            os.killpg(proc.pid, signal.SIGTERM)
            try:
                proc.wait(timeout=15)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
        except ProcessLookupError:
            pass
        rc = 124
    t.join(timeout=10)
    if held[0] is not None:
        if rc == 0:
            print(held[0], flush=True)                        # the result line: the last line of stdout
        else:
            print(held[0], file=sys.stderr, flush=True)       # a failed launch has no result
    if rc != 0:
        print(f"[launch] torch.distributed.run exited with code {rc}", file=sys.stderr, flush=True)
    return rc
