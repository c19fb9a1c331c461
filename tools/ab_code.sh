#!/bin/bash
# alternate bench runs of the working tree and of tools/_ab/<name> (tools/mkbase.sh) on ONE box:  tools/ab_code.sh [name=base] [pairs=3] [bench args...]
name=${1:-base}; n=${2:-3}
if [ $# -ge 2 ]; then shift 2; else shift $#; fi
root=$(cd "$(dirname "$0")/.." && pwd)
run() { (cd "$1" && python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernels --no-variants "${@:2}" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['median_ms_200']['median'])"); }
for i in $(seq $n); do
  echo "$name($(cat $root/tools/_ab/$name/REV | cut -c1-8))  $(run $root/tools/_ab/$name "$@")"
  echo "tree  $(run $root "$@")"
done
