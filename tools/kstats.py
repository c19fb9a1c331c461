#!/usr/bin/env python3
"""per-step view of a rocprofv3 kernel_stats.csv of bench.py: usage tools/kstats.py <csv> [steps_profiled=37] [top=45]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 37.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
tot = sum(float(r["TotalDurationNs"]) for r in rows) / n / 1e3
print(f"total kernel time per step: {tot:.0f} us")
for r in rows[:top]:
    print(f"{r['Name'][:100]:100s} {float(r['Calls'])/n:6.1f}/step  avg {float(r['AverageNs'])/1e3:7.1f} us  {float(r['TotalDurationNs'])/n/1e3:7.0f} us/step")
