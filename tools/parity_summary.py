"""Fold a parity report (gpurun_out/parity_report.txt, written by the -m gpu tests through tests.test_modules_gpu.Checks) into the
worst measured value per class of SURVEY 8c's tolerance contract -> profiles/rNN_parity_measured.json, which bench.py quotes in
`config.tolerances` beside the contract itself.

    python3 tools/parity_summary.py gpurun_out/parity_report.txt profiles/r05_parity_measured.json
"""
import json
import re
import sys

CLASSES = [          # (key, regex on the check's label, which contract bound it belongs to)
    ("fwd_rel_eval_full_batch", r"^fullsize-eval\[.*\] (pc|img) (backbone|feats) rel:"),
    ("fwd_rel_train_backbone_full_batch", r"^fullsize-train\[.*\] (pc|img) backbone rel \(fp32 oracle\):"),
    ("fwd_rel_train_feats_behind_batchnorm_full_batch", r"^fullsize-train\[.*\] (pc|img) feats rel \(fp32 oracle\):"),
    ("loss_abs_full_batch", r"^fullsize-train\[.*\] loss abs diff vs fp32 oracle"),
    ("grad_deficit_all_linear_full_batch", r"^fullsize-train\[.*\] \[linear loss\] all-parameter gradient deficit \(1 - cos\) vs fp32:"),
    ("grad_deficit_all_ntxent_full_batch", r"^fullsize-train\[.*\] \[NT-Xent loss\] all-parameter gradient deficit \(1 - cos\) vs fp32:"),
    ("grad_deficit_worst_tensor_linear_full_batch", r"^fullsize-train\[.*\] \[linear loss\] worst per-tensor gradient deficit vs fp32:"),
    ("grad_deficit_worst_tensor_ntxent_full_batch", r"^fullsize-train\[.*\] \[NT-Xent loss\] worst per-tensor gradient deficit vs fp32:"),
    # round 6: against fixtures the imported reference wrote at 64 / 32 / 16 pairs (dropout 0), tests/golden/fullsize_*.npz
    ("fwd_rel_eval_vs_reference_fixture", r"^fullsize-fixture\[.*\] (pc|img) eval (backbone|feats) rel:"),
    ("fwd_rel_train_backbone_vs_reference_fixture", r"^fullsize-fixture\[.*\] (pc|img) train backbone rel:"),
    ("fwd_rel_train_feats_behind_batchnorm_vs_reference_fixture", r"^fullsize-fixture\[.*\] (pc|img) train feats rel:"),
    ("loss_abs_vs_reference_fixture", r"^fullsize-fixture\[.*\] loss abs diff vs the reference"),
    ("grad_deficit_all_linear_vs_reference_fixture", r"^fullsize-fixture\[.*\] \[lin\] all-parameter gradient deficit"),
    ("grad_deficit_all_ntxent_vs_reference_fixture", r"^fullsize-fixture\[.*\] \[ntx\] all-parameter gradient deficit"),
    ("grad_deficit_all_ntxent_golden_batches", r"^dropout-step\[.*\] \[NT-Xent loss\] all-parameter gradient deficit \(1 - cos\) vs fp32:"),
    ("grad_deficit_worst_tensor_ntxent_golden_batches", r"^dropout-step\[.*\] \[NT-Xent loss\] worst per-tensor gradient deficit vs fp32:"),
]


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_report.txt"
    dst = sys.argv[2] if len(sys.argv) > 2 else None
    out = {}
    fails = 0
    for line in open(src).read().splitlines():
        m = re.search(r": ([0-9.e+-]+|nan|inf) \(([<>]) ([0-9.e+-]+)\)$", line)
        if not m:
            continue
        v, op, b = float(m.group(1)), m.group(2), float(m.group(3))
        if not (v < b if op == "<" else v > b):
            fails += 1
        for key, rx in CLASSES:
            if re.search(rx, line):
                e = out.setdefault(key, {"worst": 0.0, "bound_in_test": b, "checks": 0, "where": ""})
                e["checks"] += 1
                if v >= e["worst"]:
                    e["worst"], e["where"] = v, line.split("]")[0] + "]"
                e["bound_in_test"] = max(e["bound_in_test"], b)
    res = {"source": src, "failing_checks": fails, "classes": out,
           "note": "worst value per class over every architecture the -m gpu suite ran (1 - cosine for the gradient classes)"}
    txt = json.dumps(res, indent=1)
    if dst:
        open(dst, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
