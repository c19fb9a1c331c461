#!/usr/bin/env python3
"""The kernel sequence of ONE replayed step from a rocprofv3 kernel trace: start offset, duration, queue, name -- in start order.
usage: tools/step_sequence.py <kernel_trace.csv> [from_us to_us]      (a step = the launches between two adamw_kernel launches)"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adamw_kernel")]
a, b = ends[-3] + 1, ends[-2] + 1          # the second-to-last complete step
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
lo = float(sys.argv[2]) if len(sys.argv) > 2 else -1
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e12
queues = sorted({r["Queue_Id"] for r in step})
print(f"{len(step)} launches, queues {queues}, step span {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
for r in step:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if s < lo or s > hi:
        continue
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
    print(f"{s:8.1f} {e - s:7.1f} us  q{queues.index(r['Queue_Id'])}  grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):5d},{int(r['Grid_Size_Y']) // max(int(r['Workgroup_Size_Y']), 1):3d},{int(r['Grid_Size_Z']) // max(int(r['Workgroup_Size_Z']), 1):3d} x {r['Workgroup_Size_X']:>4}  {name}")
