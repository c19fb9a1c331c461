#!/bin/bash
# On the GPU box, from the repo root: what the step's kernels keep busy -- VALU / MFMA / LDS issue cycles and wave residency per kernel.
#   tools/collect_step_issue.sh r03 [arch]  -> gpurun_out/<tag>_step_issue[_arch].json  (folded by tools/step_issue.py)
# Separate --pmc passes of the SAME command (SQ counters, 4 per pass), never together with a trace domain other than --kernel-trace.
tag=${1:-r03}
arch=${2:-c2}
export TMPDIR=/tmp
out=${VPF_SCRATCH:-$PWD/gpurun_out}/stepissue_${tag}_$arch
mkdir -p $out
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pass$i -o pmc -- python3 bench.py --arch $arch --steps 6 --warmup 2 --no-cpu-baseline --no-kernels --no-variants > $out/pass$i.log 2>&1
  tail -1 $out/pass$i.log | cut -c1-100
done
sfx=""; [ "$arch" != "c2" ] && sfx="_$arch"
python3 tools/step_issue.py $PWD/gpurun_out/${tag}_step_issue$sfx.json $PWD/profiles/${tag}_step_bytes$sfx.json $(find $out -name "*counter_collection.csv") | cut -c1-220
