#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for n in 2 4 8; do
  s=$(date +%s)
  VPF_DIST_BACKEND=gloo VPF_SINGLE_GPU=1 VPF_BENCH_MEDIAN=0 VPF_BENCH_WATCHDOG_S=300 VPF_BENCH_LAUNCH_TIMEOUT_S=400 python bench.py --gpus $n --steps 5 --warmup 2 --pairs 8 --no-cpu-baseline --no-kernels > gpurun_out/selflaunch_$n.json 2> gpurun_out/selflaunch_$n.err
  rc=$?
  echo "self-launched --gpus $n on one GPU over gloo: rc $rc, $(( $(date +%s) - s )) s, stdout lines $(wc -l < gpurun_out/selflaunch_$n.json)" | tee -a gpurun_out/r06_selflaunch_one_gpu.txt
  python3 -c "
import json
d=json.loads(open('gpurun_out/selflaunch_$n.json').read().strip().splitlines()[-1])
c=d['config']
print('  n_gpus', d['n_gpus'], 'ranks_seen', c['ranks_seen'], 'capture', c['capture'], 'hip_graph', c['hip_graph'], 'global_batch', c['global_batch'], 'value', d['value'], 'ms/step', d['ms_per_step'], 'comm_ms', c['comm_ms'], 'losses_finite', c['losses_finite'], 'variants', sorted(d.get('variants', {}).keys())[:3])
" | tee -a gpurun_out/r06_selflaunch_one_gpu.txt
done
