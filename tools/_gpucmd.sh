#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { (cd "$1" && env $3 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernels --no-variants $2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['median_ms_200']['median'])"); }
{
for arch in c2 ref144; do
  for i in 1 2 3; do
    echo "$arch r04head        $(run tools/_ab/r04head "--arch $arch" X=1)"
    echo "$arch tree DMA=0     $(run . "--arch $arch" VPF_WGROUP_DMA=0)"
    echo "$arch tree default   $(run . "--arch $arch" X=1)"
  done
done
} > gpurun_out/r05_regress2.txt 2>&1
cat gpurun_out/r05_regress2.txt | grep -v amdgpu.ids
