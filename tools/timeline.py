#!/usr/bin/env python3
"""coarse per-queue timeline of the last full step in a rocprofv3 kernel trace of bench.py: tools/timeline.py <trace.csv> [window_us=400]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
W = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 400e3
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows)
starts = [i for i, k in enumerate(ks) if "fps_kernel" in k[3]]
seg = ks[starts[-2]:starts[-1]]
t0 = seg[0][0]
n = int((seg[-1][1] - t0) / W) + 1
for w in range(n):
    a = t0 + w * W; b = a + W
    d = collections.defaultdict(float); names = collections.defaultdict(collections.Counter)
    for s, e, q, nm in seg:
        o = max(0, min(e, b) - max(s, a))
        if o > 0:
            d[q] += o; names[q][nm.split('(')[0].replace('void ', '')[:28]] += o
    line = f"{w*W/1e3:6.0f}us: "
    for q in sorted(d):
        top = ", ".join(f"{k}:{v/1e3:.0f}" for k, v in names[q].most_common(3))
        line += f" q{q} {d[q]/1e3:4.0f} [{top}]"
    print(line[:250])
