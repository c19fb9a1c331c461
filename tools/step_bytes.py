#!/usr/bin/env python3
"""Whole-step HBM byte budget: fold one rocprofv3 --pmc FETCH_SIZE pass, one --pmc WRITE_SIZE pass and the --kernel-trace --stats
summary of the SAME command (python3 bench.py --no-kernels --no-cpu-baseline --no-variants) into profiles/<round>_step_bytes.json.

    tools/step_bytes.py <out.json> <kernel_trace.csv> <fetch_counter_collection.csv> <write_counter_collection.csv> [ms_per_step]

Per kernel (template arguments kept, parameter lists dropped): launches per step, in-step average microseconds, HBM bytes per
launch and per step.  bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): MI355X_MICROARCH.md section HBM -- gfx950 tallies a wide
coalesced read at half its bytes; WRITE_SIZE is exact for 16-byte stores and float atomics; other widths are uncalibrated, so the
absolute figure is an estimate and the ranking is what it is for.  Only the last whole steps of each run count (delimited by adamw_kernel, one launch per step).
"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in name:                     # drop the (parameter list) but keep <template arguments>
        if ch == "(" and depth == 0:
            break
        out.append(ch)
        depth += (ch == "<") - (ch == ">")
    return "".join(out).strip()


def windows(rows, name_key, last=5):
    """Dispatch rows of one run ordered by start time -> the rows of the LAST `last` whole steps.  A step = what lies between two
    consecutive adamw_kernel dispatches (one per step, single rank): initialisation, capture warm-ups and the first replays fall away."""
    rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if short(r[name_key]).startswith("adamw_kernel")]
    if len(marks) < 2:
        return rows, 1
    k = min(last, len(marks) - 1)
    return rows[marks[-1 - k] + 1:marks[-1] + 1], k


def load_counter(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows, steps = windows(rows, "Kernel_Name")
    agg = collections.defaultdict(list)
    for r in rows:
        agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg, steps


def load_trace(path):
    rows, steps = windows(list(csv.DictReader(open(path))), "Kernel_Name")
    st = collections.OrderedDict()
    for r in rows:
        e = st.setdefault(short(r["Kernel_Name"]), dict(calls=0.0, ns=0.0))
        e["calls"] += 1.0; e["ns"] += float(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return st, steps


def main():
    out, trace, fcsv, wcsv = sys.argv[1:5]
    ms_step = float(sys.argv[5]) if len(sys.argv) > 5 else None
    (fetch, steps_f), (write, steps_w) = load_counter(fcsv, "FETCH_SIZE"), load_counter(wcsv, "WRITE_SIZE")
    st, steps_t = load_trace(trace)
    kernels = []
    tot_b = tot_us = tot_f = tot_w = 0.0
    names = list(st) + [k for k in fetch if k not in st]
    for k in names:
        f, w = fetch.get(k, []), write.get(k, [])
        fb = 2.0 * 1024.0 * sum(f) / steps_f            # bytes per step, read side (x2: the guide's gfx950 correction)
        wb = 1024.0 * sum(w) / steps_w
        n = st[k]["calls"] / steps_t if k in st else len(f) / steps_f
        us = st[k]["ns"] / steps_t / 1e3 if k in st else 0.0
        e = dict(kernel=k, launches_per_step=round(n, 2), us_per_step=round(us, 1),
                 avg_us_in_step=round(us / n, 2) if n else None,
                 fetch_bytes_per_step=round(fb), write_bytes_per_step=round(wb), hbm_bytes_per_step=round(fb + wb),
                 hbm_bytes_per_launch=round((fb + wb) / n) if n else None,
                 tb_per_s_in_step=round((fb + wb) / us / 1e6, 3) if us else None)
        kernels.append(e)
        tot_b += fb + wb; tot_us += us; tot_f += fb; tot_w += wb
    kernels.sort(key=lambda e: -e["hbm_bytes_per_step"])
    res = dict(command="python3 bench.py --no-kernels --no-cpu-baseline --no-variants (rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | "
                       "--kernel-trace: three separate runs; the last whole steps of each, delimited by adamw_kernel)",
               steps_in_fetch_pass=steps_f, steps_in_write_pass=steps_w, steps_in_trace=steps_t,
               kernels_per_step=round(sum(e["calls"] for e in st.values()) / steps_t, 1),
               kernel_us_per_step=round(tot_us, 1), hbm_bytes_per_step=round(tot_b), fetch_bytes_per_step=round(tot_f),
               write_bytes_per_step=round(tot_w), kernels=kernels)
    if ms_step:
        res["ms_per_step_unprofiled"] = ms_step
        res["step_tb_per_s"] = round(tot_b / (ms_step * 1e-3) / 1e12, 3)
        res["step_hbm_frac_of_8tbs"] = round(tot_b / (ms_step * 1e-3) / 8e12, 4)
    try:            # the library build the budget belongs to: bench.py drops a budget whose id is not the loaded library's
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from vipformer_amd import build as _b
        res["build_id"] = _b.built_hash()
    except Exception as e:                                  # noqa: BLE001
        res["build_id"] = None
        print("no build id:", e)
    json.dump(res, open(out, "w"), indent=1)
    print(f"steps f/w/t {steps_f}/{steps_w}/{steps_t}  kernels/step {res['kernels_per_step']}  kernel us/step {tot_us:.0f}  "
          f"HBM GB/step {tot_b / 1e9:.2f} (fetch {tot_f / 1e9:.2f} + write {tot_w / 1e9:.2f})"
          + (f"  = {res['step_tb_per_s']} TB/s = {res['step_hbm_frac_of_8tbs']} of 8 TB/s" if ms_step else ""))
    for e in kernels[:45]:
        print(f"{e['kernel'][:70]:70s} {e['launches_per_step']:6.1f}/step {e['us_per_step']:8.1f} us  {e['hbm_bytes_per_step'] / 1e6:9.1f} MB/step"
              f"  {(e['hbm_bytes_per_launch'] or 0) / 1e6:8.1f} MB/launch  {e['tb_per_s_in_step'] or 0:6.2f} TB/s")


if __name__ == "__main__":
    main()
