#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 1 2; do
  out=/tmp/kp$v; rm -rf $out; mkdir -p $out
  VPF_KNN_SELECT=$v rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- python3 tools/microbench.py preproc > $out/log.txt 2>&1
  python3 - $(find $out -name "*kernel_stats.csv" | head -1) $v <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "knn_group" in r["Name"] or "fps_kernel" in r["Name"]:
        print(f"VPF_KNN_SELECT={sys.argv[2]}  {r['Name'][5:45]:42s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:7.1f} us  min {float(r['MinNs'])/1e3:7.1f}")
PY
done
