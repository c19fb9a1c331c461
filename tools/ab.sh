#!/bin/bash
# A/B of one environment switch on ONE box, alternating runs:  tools/ab.sh "VPF_SA_WG2=0" "VPF_SA_WG2=1" [pairs=3] [bench args...]
a=$1; b=$2; n=3
if [[ "${3:-}" =~ ^[0-9]+$ ]]; then n=$3; shift 3; else shift 2; fi
for i in $(seq $n); do
  for cfg in "$a" "$b"; do
    ms=$(env $cfg python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernels --no-variants "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$cfg  $ms ms/step"
  done
done
