#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# (1) the multi-process file 25 x under pytest (the driver's conditions: the pytest process holds a HIP context and has had an RCCL group)
rm -f gpurun_out/dp2_test_failure.txt
for i in $(seq 1 25); do
  s=$(date +%s)
  DP2_WATCHDOG_S=60 timeout 600 python -m pytest tests/test_zz_multiproc_gpu.py -x -q -m gpu > gpurun_out/zz_iter.log 2>&1
  rc=$?
  echo "zz iteration $i rc $rc $(( $(date +%s) - s )) s: $(tail -1 gpurun_out/zz_iter.log)" >> gpurun_out/zz_loop_r06.log
  if [ $rc -ne 0 ]; then cp gpurun_out/zz_iter.log gpurun_out/zz_fail_$i.log; fi
done
tail -30 gpurun_out/zz_loop_r06.log
# (2) packed-fp32 A/B: base vs encoder kernels with SLP vectorisation (pkf32e) vs + attention (pkf32)
for i in 1 2 3; do
  for v in base pkf32e pkf32; do
    if [ $v = base ]; then unset VPF_LIB; else export VPF_LIB=$PWD/tools/_bin/libvipformer_$v.so; fi
    ms=$(python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernels --no-variants 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['median_ms_200']['median'])")
    echo "$v $ms" | tee -a gpurun_out/pkf32_ab_r06.txt
  done
done
unset VPF_LIB
