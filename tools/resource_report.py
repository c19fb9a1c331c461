#!/usr/bin/env python3
"""Per-kernel resource report (VGPRs, AGPRs, scratch, LDS, occupancy) of one csrc/*.hip file, from the compiler's
-Rpass-analysis=kernel-resource-usage remarks.  rocprofv3's VGPR_Count column is not the allocation;
this is.   usage: tools/resource_report.py sa_layer.hip [substring-filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vipformer_amd import build as B  # noqa: E402


def report(fname: str, flt: str = "") -> list:
    src = os.path.join(B.CSRC, fname)
    cmd = ["hipcc"] + B.COMMON + B.PER_FILE.get(fname, []) + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        txt = m.group(1).strip()
        if txt.startswith("Function Name:") or txt.startswith("Name:"):
            name = txt.split(":", 1)[1].strip()
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            cur = {"name": name}
            rows.append(cur)
        elif cur is not None and ":" in txt:
            k, v = txt.split(":", 1)
            cur[k.strip()] = v.strip()
    return [r for r in rows if flt in r["name"]]


if __name__ == "__main__":
    f = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    print(f"{'kernel':<72} {'VGPR':>5} {'AGPR':>5} {'scratch':>8} {'LDS':>7} {'occ':>4} {'SGPR':>5}")
    for r in report(f, flt):
        nm = re.sub(r"\(anonymous namespace\)::", "", r["name"]); nm = re.sub(r"^void ", "", nm); nm = re.sub(r"\(.*\)$", "", nm)[:72]
        print(f"{nm:<72} {r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('ScratchSize [bytes/lane]', '?'):>8} "
              f"{r.get('LDS Size [bytes/block]', '?'):>7} {r.get('Occupancy [waves/SIMD]', '?'):>4} {r.get('TotalSGPRs', '?'):>5}")
