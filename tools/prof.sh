#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args]
# kernel-trace + stats of bench.py; summary copied to gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 bench.py --no-cpu-baseline "$@" > $out/bench.log 2>&1
tail -2 $out/bench.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $PWD/gpurun_out/${tag}_kernel_stats.csv 2>/dev/null
head -40 "$f" | cut -c1-200
