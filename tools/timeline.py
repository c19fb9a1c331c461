#!/usr/bin/env python3
"""One steady-state step of a rocprofv3 --kernel-trace as a timeline: every dispatch with its start (us from the step's first
kernel), duration, queue, and the number of kernels running at its start; then the busy / idle summary.

    tools/timeline.py <kernel_trace.csv> [step_from_the_end=2] [--gaps-only]

A step = the dispatches between two adamw_kernel launches.  "idle" = no kernel of this process running; "solo" = exactly one.
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in name:
        if ch == "(" and depth == 0:
            break
        out.append(ch)
        depth += (ch == "<") - (ch == ">")
    return "".join(out).strip()[:58]


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    back = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 2
    marks = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith("adamw_kernel")]
    step = rows[marks[-1 - back] + 1:marks[-back] + 1]
    t0 = int(step[0]["Start_Timestamp"])
    ev = [((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), short(r["Kernel_Name"])) for r in step]
    queues = {q: i for i, q in enumerate(sorted({e[2] for e in ev}))}
    end = max(e[1] for e in ev)
    # sweep
    pts = sorted([(s, 1) for s, _, _, _ in ev] + [(e, -1) for _, e, _, _ in ev])
    busy = {0: 0.0, 1: 0.0, 2: 0.0}
    n, last = 0, 0.0
    gaps = []
    for t, d in pts:
        busy[min(n, 2)] += t - last
        if n == 0 and t - last > 1.0:
            gaps.append((last, t))
        last = t; n += d
    if "--gaps-only" not in sys.argv:
        for s, e, q, k in ev:
            conc = sum(1 for s2, e2, _, _ in ev if s2 <= s < e2) - 1
            print(f"{s:8.1f} {e - s:7.1f}  q{queues[q]}  +{conc}  {k}")
    print(f"# step {end:.1f} us: idle {busy[0]:.1f}  solo {busy[1]:.1f}  overlapped {busy[2]:.1f}; {len(ev)} dispatches, sum of durations {sum(e - s for s, e, _, _ in ev):.1f}")
    print("# idle gaps > 1 us:", " ".join(f"{a:.0f}-{b:.0f}" for a, b in gaps))


if __name__ == "__main__":
    main()
