#!/bin/bash
# round 5: store cache policy of the row-block kernel; tile shapes of the single weight-gradient GEMMs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== bitwise"; timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "sa_rows_fwd" 2>&1 | tail -5
for pol in 0 1 2 3; do
  echo "== DEC store policy $pol"; VPF_SA_WG2=1 VPF_SA_RB=13 VPF_SA_STORE=$pol timeout 300 python3 tools/microbench.py satail3 satail 2>&1 | grep -v amdgpu.ids
done
echo "== A/B wgrad cfg 0 vs 2"; bash tools/ab.sh "VPF_WGRAD_CFG=0" "VPF_WGRAD_CFG=2" 2
echo "== A/B wgrad cfg 0 vs 1"; bash tools/ab.sh "VPF_WGRAD_CFG=0" "VPF_WGRAD_CFG=1" 2
} > gpurun_out/r05_store.txt 2>&1
tail -80 gpurun_out/r05_store.txt
