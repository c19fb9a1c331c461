#!/usr/bin/env python3
"""What each kernel of the step keeps busy: fold the SQ counter passes of tools/collect_step_issue.sh into one table.

    tools/step_issue.py <out.json> <step_bytes.json (durations, launches)> <counter_collection.csv> ...

Per kernel and per step (the last whole steps of each pass, delimited by adamw_kernel as in tools/step_bytes.py):
  valu / mfma / lds  = cycles the unit executes, as a fraction of the SIMD-cycles (4 x 256 SIMDs x duration; LDS: 256 CUs) the kernel
                       had the chip for (its in-step duration from the committed kernel trace, 2.4 GHz);
                       SQ_ACTIVE_INST_VALU and SQ_ACTIVE_INST_LDS count 4-cycle issue quads, SQ_VALU_MFMA_BUSY_CYCLES cycles
                       (checked on sa_layer_fwd: 393 216 MFMAs x 32 cycles = 12 582 912 = the counter);
  wait               = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: share of its resident time a wave sits in s_waitcnt;
  valu_per_mfma      = VALU instructions issued per MFMA instruction (SQ_INSTS_VALU counts the MFMAs too: subtracted).
"""
import collections
import csv
import json
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from step_bytes import short, windows  # noqa: E402

CLK = 2.4e9


def main():
    out, bytes_json = sys.argv[1], sys.argv[2]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    for path in sys.argv[3:]:
        rows = list(csv.DictReader(open(path)))
        for c in sorted({r["Counter_Name"] for r in rows}):
            sel, steps = windows([r for r in rows if r["Counter_Name"] == c], "Kernel_Name")
            for r in sel:
                tot[short(r["Kernel_Name"])][c] += float(r["Counter_Value"]) / steps
    prof = {e["kernel"]: e for e in json.load(open(bytes_json))["kernels"]}
    res = []
    for k, c in tot.items():
        p = prof.get(k)
        if not p or not p["us_per_step"]:
            continue
        simd_cycles = p["us_per_step"] * 1e-6 * CLK * 1024.0
        mf = c.get("SQ_INSTS_MFMA", 0.0)
        e = dict(kernel=k, launches_per_step=p["launches_per_step"], us_per_step=p["us_per_step"],
                 valu=round(4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles, 4),
                 mfma=round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles, 4),
                 lds=round(4.0 * c.get("SQ_ACTIVE_INST_LDS", 0.0) / (simd_cycles / 4.0), 4),
                 wait=round(c.get("SQ_WAIT_INST_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0), 4),
                 waves_resident_per_simd=round(4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / simd_cycles, 2),
                 valu_per_mfma=round((c.get("SQ_INSTS_VALU", 0.0) - mf) / mf, 1) if mf else None,
                 insts={n[9:].lower(): round(v) for n, v in c.items() if n.startswith("SQ_INSTS_")},
                 lds_conflict=round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 3))
        res.append(e)
    res.sort(key=lambda e: -e["us_per_step"])
    T = sum(e["us_per_step"] for e in res)
    agg = {u: round(sum(e[u] * e["us_per_step"] for e in res) / T, 4) for u in ("valu", "mfma", "lds", "wait")}
    json.dump(dict(note=__doc__.split("\n\n")[2], kernel_time_weighted=agg, kernels=res), open(out, "w"), indent=1)
    print(f"kernel-time-weighted over {T:.0f} us/step: " + "  ".join(f"{u} {v:.3f}" for u, v in agg.items()))
    print(f"{'kernel':62s} {'n':>4s} {'us/step':>8s} {'valu':>6s} {'mfma':>6s} {'lds':>6s} {'wait':>6s} {'waves':>6s} {'valu/mfma':>9s} {'ldsconf':>7s}")
    for e in res[:40]:
        print(f"{e['kernel'][:62]:62s} {e['launches_per_step']:4.0f} {e['us_per_step']:8.1f} {e['valu']:6.3f} {e['mfma']:6.3f} {e['lds']:6.3f} {e['wait']:6.3f} "
              f"{e['waves_resident_per_simd']:6.2f} {str(e['valu_per_mfma']):>9s} {e['lds_conflict']:7.3f}")


if __name__ == "__main__":
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    main()
