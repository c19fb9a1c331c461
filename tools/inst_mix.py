#!/usr/bin/env python3
"""Static instruction mix of one kernel of the built library: MFMA / VALU / transcendental / LDS / VMEM / SALU / waitcnt / barrier
counts from llvm-objdump's disassembly of the gfx950 code object, plus the backward branches (loops) with the counts inside each
loop body.   usage: tools/inst_mix.py <kernel-name-substring> [lib.so]

Static counts are dynamic counts only for straight-line code: a loop body's counts must be multiplied by its trip count by hand
(the report lists each loop's span so that is possible)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vipformer_amd import build as B  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_accvgpr"):
        return "accmov"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernels(lib):
    for img in B.device_code_objects(lib):
        path = "/tmp/_inst_mix.co"
        open(path, "wb").write(img)
        txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
        cur, body = None, []
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                if cur:
                    yield cur, body
                cur, body = m.group(1), []
            elif cur and line.strip():
                body.append(line)
        if cur:
            yield cur, body


def main():
    want = sys.argv[1]
    lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "vipformer_amd", "libvipformer_hip.so")
    for name, body in kernels(lib):
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if want not in dem and want not in name:
            continue
        insts = []                     # (address, opcode, operands)
        for line in body:
            m = re.match(r"^\s*(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", line)
            if m:
                insts.append((int(m.group(3), 16), m.group(1), m.group(2) + " " + m.group(4)))
        addr_index = {a: i for i, (a, _, _) in enumerate(insts)}
        total = collections.Counter(classify(op) for _, op, _ in insts)
        print(f"== {dem[:150]}\n   {len(insts)} instructions: " + "  ".join(f"{k} {v}" for k, v in sorted(total.items(), key=lambda kv: -kv[1])))
        # loops: backward branches
        for i, (a, op, args) in enumerate(insts):
            if classify(op) == "branch":
                tgt = None
                m2 = re.search(r"\+0x([0-9a-fA-F]+)>", args)
                if m2:
                    tgt = insts[0][0] + int(m2.group(1), 16)
                if tgt is not None and tgt <= a and tgt in addr_index:
                    j = addr_index[tgt]
                    c = collections.Counter(classify(o) for _, o, _ in insts[j:i + 1])
                    print(f"   loop [{j}..{i}] ({i - j + 1} insts): " + "  ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
