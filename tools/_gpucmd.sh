#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export VPF_SCRATCH=/tmp/vpf_prof; mkdir -p $VPF_SCRATCH
rm -f gpurun_out/parity_report.txt
s=$(date +%s)
python -m pytest tests/ -x -q -m gpu > gpurun_out/final_suite.log 2>&1
echo "final tree: rc $? $(( $(date +%s) - s )) s: $(grep -E "passed|failed" gpurun_out/final_suite.log | tail -1)" | tee gpurun_out/final_suite_summary.txt
cp gpurun_out/parity_report.txt gpurun_out/r06_parity_report.txt
bash tools/collect_step_bytes.sh r06 c2 2>&1 | tail -1
cp gpurun_out/r06_step_bytes.json profiles/r06_step_bytes.json
bash tools/collect_step_issue.sh r06 c2 2>&1 | tail -1
bash tools/collect_profiles.sh r06 2>&1 | tail -2
for a in c3 c4 ref144 ref144m4; do
  bash tools/collect_step_bytes.sh r06 $a 2>&1 | tail -1
  cp gpurun_out/r06_step_bytes_$a.json profiles/
  python3 bench.py --arch $a --no-cpu-baseline 2> gpurun_out/r06_bench_$a.err | tail -1 > gpurun_out/r06_bench_$a.json
done
python3 bench.py --arch c5 2> gpurun_out/r06_bench_c5.err | tail -1 > gpurun_out/r06_bench_c5.json
python3 bench.py 2> gpurun_out/r06_bench_default.err | tail -1 > gpurun_out/r06_bench_default.json
for f in default c3 c4 ref144 ref144m4 c5; do python3 -c "import json;d=json.load(open('gpurun_out/r06_bench_$f.json'));print('$f',d['value'],d['ms_per_step'],(d.get('roofline') or {}).get('frac'),(d.get('roofline') or {}).get('frac_is'))"; done
