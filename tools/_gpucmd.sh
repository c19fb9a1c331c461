#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== tests"; timeout 1500 python3 -m pytest tests/test_modules_gpu.py tests/test_kernels_gpu.py -x -q -k "stages_vs_reference_golden or models_vs_reference_golden or group2emb or g2e or training_step_with_dropout" 2>&1 | tail -5
bash tools/kprof.sh g2e3 "g2e" X=1 -- g2e > /dev/null 2>&1
f=$(find gpurun_out/kprof_g2e3 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'g2e' in r['Name'] and float(r['AverageNs'])>20000: print(r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1), r['Name'][:60])
PY
timeout 300 python3 tools/microbench.py g2e 2>&1 | grep "pass "
echo "== code A/B: base2 vs tree"; bash tools/ab_code.sh base2 3
} > gpurun_out/r05_p0dma.txt 2>&1
cat gpurun_out/r05_p0dma.txt | grep -v amdgpu.ids
