#!/bin/bash
# On the GPU box, from the repo root: the whole-step byte budget.
#   tools/collect_step_bytes.sh r03 [arch]   -> gpurun_out/<tag>_bench_<arch>_kernel_stats.csv, gpurun_out/<tag>_step_bytes[_arch].json
# Three runs of the SAME command: kernel trace + statistics, --pmc FETCH_SIZE, --pmc WRITE_SIZE (the TCC slot budget does not fit both
# in one pass; counters never together with a trace domain other than --kernel-trace).  The program follows `--` directly.
tag=${1:-r03}
arch=${2:-c2}
export TMPDIR=/tmp
out=${VPF_SCRATCH:-$PWD/gpurun_out}/stepbytes_${tag}_$arch
mkdir -p $out
python3 bench.py --arch $arch --steps 30 --warmup 5 --no-cpu-baseline --no-kernels --no-variants > $out/plain.log 2>&1
ms=$(tail -1 $out/plain.log | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
echo "unprofiled ms/step: $ms"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 bench.py --arch $arch --steps 30 --warmup 5 --no-cpu-baseline --no-kernels --no-variants > $out/trace.log 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $PWD/gpurun_out/${tag}_bench_${arch}_kernel_stats.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$ctr -o pmc -- python3 bench.py --arch $arch --steps 6 --warmup 2 --no-cpu-baseline --no-kernels --no-variants > $out/pmc_$ctr.log 2>&1
  tail -1 $out/pmc_$ctr.log | cut -c1-120
done
sfx=""; [ "$arch" != "c2" ] && sfx="_$arch"
python3 tools/step_bytes.py $PWD/gpurun_out/${tag}_step_bytes$sfx.json $(find $out/trace -name "*kernel_trace.csv" | head -1) \
  $(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $ms | cut -c1-200
