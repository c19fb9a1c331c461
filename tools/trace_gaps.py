#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 kernel trace of bench.py: for the last replayed step, per queue: busy time, gaps,
and the union (wall) -- tells how much of the step is kernel time vs dependency/launch gaps.
usage: tools/trace_gaps.py <kernel_trace.csv> [n_last_steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows]
ks.sort()
# a step starts at each fps_kernel
starts = [i for i, k in enumerate(ks) if "fps_kernel" in k[3]]
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for si in range(len(starts) - nl - 1, len(starts) - 1):
    seg = ks[starts[si]:starts[si + 1]]
    t0, t1 = seg[0][0], max(k[1] for k in seg)
    perq = collections.defaultdict(list)
    for k in seg: perq[k[2]].append(k)
    # union busy
    ev = sorted((k[0], k[1]) for k in seg)
    busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    for s, e in ev[1:]:
        if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    msg = f"step {si}: wall {(ks[starts[si+1]][0]-t0)/1e3:.0f} us, kernels {len(seg)}, sum-dur {sum(k[1]-k[0] for k in seg)/1e3:.0f}, union-busy {busy/1e3:.0f}"
    for q, l in perq.items():
        d = sum(k[1] - k[0] for k in l)
        span = max(k[1] for k in l) - l[0][0]
        msg += f" | q{q}: n={len(l)} dur={d/1e3:.0f} span={span/1e3:.0f}"
    print(msg)
if len(sys.argv) > 3:
    seg = ks[starts[-2]:starts[-1]]
    prev_end = {}
    for k in seg:
        gap = (k[0] - prev_end.get(k[2], k[0])) / 1e3
        print(f"{(k[0]-seg[0][0])/1e3:9.1f} q{k[2]} gap {gap:6.1f} dur {(k[1]-k[0])/1e3:7.1f}  {k[3][:90]}")
        prev_end[k[2]] = k[1]
