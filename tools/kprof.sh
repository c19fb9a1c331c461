#!/bin/bash
# per-dispatch kernel durations of a microbench target under rocprofv3:  tools/kprof.sh <tag> <name-substring> <env...> -- <microbench targets>
tag=$1; sub=$2; shift 2
envs=()
while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
export TMPDIR=/tmp
out=${VPF_SCRATCH:-$PWD/gpurun_out}/kprof_$tag
rm -rf $out; mkdir -p $out
for e in "${envs[@]}"; do export "$e"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- python3 tools/microbench.py "$@" > $out/log.txt 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
grep "$sub" "$f" | cut -d, -f1-8 | cut -c1-200
