#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp VPF_SCRATCH=/tmp/vpf_prof
mkdir -p $VPF_SCRATCH
for v in base poly; do
  if [ $v = poly ]; then L="VPF_LIB=$PWD/tools/_bin/libvipformer_gelupoly.so"; else L="VPF_NOP=1"; fi
  echo "== satail3 $v"; bash tools/kprof.sh s3$v sa_layer_fwd "$L" SATAIL_B=128 -- satail3
done
for i in 1 2 3; do
  for v in base poly; do
    if [ $v = poly ]; then export VPF_LIB=$PWD/tools/_bin/libvipformer_gelupoly.so; else unset VPF_LIB; fi
    ms=$(python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernels --no-variants 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['median_ms_200']['median'])")
    echo "$v $ms"
  done
done
