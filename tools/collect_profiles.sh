#!/bin/bash
# On the GPU box, from the repo root: the rocprofv3 evidence behind bench.py's roofline / kernels entries.
#   tools/collect_profiles.sh r02        -> gpurun_out/r02_bench_c2_kernel_stats.csv, gpurun_out/r02_pmc_summary.json (+ raw CSVs)
# (1) kernel trace + statistics of the bench command itself; (2) PMC passes in their OWN runs (--pmc with --kernel-trace only: the
# guide's HBM recipe -- FETCH_SIZE and WRITE_SIZE cannot share a pass) over the stand-alone kernel harness (tools/microbench.py).
tag=${1:-r02}
export TMPDIR=/tmp
out=${VPF_SCRATCH:-$PWD/gpurun_out}/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernels > $out/bench.log 2>&1
tail -1 $out/bench.log | cut -c1-200
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $PWD/gpurun_out/${tag}_bench_c2_kernel_stats.csv
csvs=""
for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  d=$out/pmc_$(echo $ctr | cut -d' ' -f1)
  mkdir -p $d
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -o pmc -- python3 tools/microbench.py preproc attn wgroup satail > $d/log.txt 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  csvs="$csvs $f"
done
python3 tools/pmc_summary.py $PWD/gpurun_out/${tag}_pmc_summary.json $csvs | cut -c1-260
