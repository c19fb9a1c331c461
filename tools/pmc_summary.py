#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs (one pass per counter set, MI355X_MICROARCH.md "rocprofv3 PMC slots") into
profiles/<round>_pmc_summary.json, keyed by the kernel legs of bench.py.

    tools/pmc_summary.py <out.json> <csv> [<csv> ...]

Per kernel (name prefix + launch grid) the MEDIAN of every counter over its dispatches; then
  hbm_bytes_per_launch = 2 * FETCH_SIZE + WRITE_SIZE   (KiB -> bytes; gfx950 tallies a wide coalesced read at half its bytes)
  mfma_busy_frac       = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)
                         (SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles per v_mfma_f32_32x32x16_f16 summed over every SIMD -- checked against
                         the algorithmic MFMA count of the grouped weight gradient: 12 582 912 = 32 x 393 216; GRBM_GUI_ACTIVE is summed
                         over the 8 XCDs: 1.25 M for a 59 us kernel at 2.4 GHz)
  mfma_busy_over_cu_busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES   (as recorded; units of the latter uncalibrated)
"""
import collections
import csv
import json
import sys

LEGS = [  # (leg key of bench.py, kernel-name substring, predicate on (grid, first dispatch index) or None)
    ("fps_kernel", "fps_kernel", None),
    ("knn_group_select_kernel", "knn_group_select_kernel", None),
    ("gemm_wgrad_group_kernel", "gemm_wgrad_group_kernel", None),
    ("sa_layer_fwd_kernel", "sa_layer_fwd_kernel", None),
    ("attn_fwd_kernel (cross-attention pc)", "attn_fwd_kernel", None),
    ("attn_res_fwd_kernel (self-attention pc)", "attn_res_fwd_kernel", "small"),
    ("attn_res_fwd_kernel (self-attention img)", "attn_res_fwd_kernel", "large"),
    ("attn_res_bwd_kernel (self-attention pc)", "attn_res_bwd_kernel", "small"),
    ("attn_res_bwd_kernel (self-attention img)", "attn_res_bwd_kernel", "large"),
    ("attn_bwd_dq/dkv_kernel (cross-attention pc)", "attn_bwd_", None),      # dq + dkv, or the one attn_bwd_ca_kernel
]


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def main():
    out, paths = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))      # (kernel, grid) -> counter -> values
    for p in paths:
        for r in csv.DictReader(open(p)):
            grid = r.get("Grid_Size") or "x".join(r.get(k, "") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
            agg[(r["Kernel_Name"], grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for key, sub, which in LEGS:
        cands = [(k, d) for k, d in agg.items() if sub in k[0] and max(len(v) for v in d.values()) >= 3]
        if not cands:
            continue
        if sub == "attn_bwd_":           # two kernels (dq, dkv) per backward: sum them (one when attn_bwd_ca_kernel runs)
            ctrs = collections.defaultdict(float)
            for k, d in cands:
                for c, v in d.items():
                    ctrs[c] += med(v)
        else:
            def gsize(k):
                try:
                    return int(k[1].split("x")[0])
                except ValueError:
                    return 0
            cands.sort(key=lambda kd: gsize(kd[0]))
            k, d = cands[0] if which == "small" else cands[-1]
            ctrs = {c: med(v) for c, v in d.items()}
        e = {"pmc": {c: round(v, 1) for c, v in ctrs.items()}}
        if "FETCH_SIZE" in ctrs and "WRITE_SIZE" in ctrs:
            e["hbm_bytes_per_launch"] = (2.0 * ctrs["FETCH_SIZE"] + ctrs["WRITE_SIZE"]) * 1024.0
            e["hbm_fetch_bytes_x2"] = 2.0 * ctrs["FETCH_SIZE"] * 1024.0
            e["hbm_write_bytes"] = ctrs["WRITE_SIZE"] * 1024.0
        if "SQ_VALU_MFMA_BUSY_CYCLES" in ctrs and ctrs.get("GRBM_GUI_ACTIVE"):
            e["mfma_busy_frac"] = round(ctrs["SQ_VALU_MFMA_BUSY_CYCLES"] / (ctrs["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4), 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in ctrs and ctrs.get("SQ_BUSY_CU_CYCLES"):
            e["mfma_busy_over_cu_busy"] = round(ctrs["SQ_VALU_MFMA_BUSY_CYCLES"] / ctrs["SQ_BUSY_CU_CYCLES"], 4)
        res[key] = e
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, e in res.items():
        print(k, {a: b for a, b in e.items() if a != "pmc"}, e["pmc"])


if __name__ == "__main__":
    main()
