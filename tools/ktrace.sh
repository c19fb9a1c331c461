#!/bin/bash
# usage: tools/ktrace.sh <tag> <microbench targets...>   -> per-kernel device durations (rocprofv3 kernel trace)
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/kt_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 tools/microbench.py "$@" > $out/log.txt 2>&1
grep -v "^W2026\|^E2026\|amdgpu.ids" $out/log.txt | tail -20
python3 - "$out/${tag}_kernel_trace.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# consecutive runs of the same kernel+grid -> one line
prev = None; acc = []
def flush():
    if prev and len(acc) >= 5:
        a = sorted(acc); print(f"{a[len(a)//2]/1e3:9.1f} us (n={len(a)})  grid={prev[1]} lds={prev[2]}  {prev[0][:100]}")
for r in rows:
    key = (r['Kernel_Name'], r.get('Grid_Size_X', '') + 'x' + r.get('Grid_Size_Y', '') + 'x' + r.get('Grid_Size_Z', ''), r.get('LDS_Block_Size', ''))
    d = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    if key != prev:
        flush(); prev = key; acc = []
    acc.append(d)
flush()
PY
