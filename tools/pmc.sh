#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" <microbench target>
tag=$1; ctr=$2; shift 2
export TMPDIR=/tmp
out=${VPF_SCRATCH:-$PWD/gpurun_out}/pmc_$tag
mkdir -p $out
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o $tag -- python3 tools/microbench.py "$@" > $out/log.txt 2>&1
tail -3 $out/log.txt
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
