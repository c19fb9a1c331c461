#!/bin/bash
# round 5 profile pipeline (both parts) at the final build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp VPF_SCRATCH=/tmp/vpf_prof
mkdir -p $VPF_SCRATCH
{
bash tools/collect_step_bytes.sh r05 c2
cp gpurun_out/r05_step_bytes.json profiles/r05_step_bytes.json
bash tools/collect_step_issue.sh r05 c2
bash tools/collect_profiles.sh r05
cp gpurun_out/r05_pmc_summary.json profiles/r05_pmc_summary.json
python3 tools/step_timeline.py > gpurun_out/r05_step_timeline_c2.txt 2>/dev/null
for arch in c3 c4 ref144 ref144m4; do
  bash tools/collect_step_bytes.sh r05 $arch > /dev/null 2>&1
  cp gpurun_out/r05_step_bytes_$arch.json profiles/ 2>/dev/null
  python3 bench.py --arch $arch --no-cpu-baseline --no-variants > gpurun_out/r05_bench_$arch.json 2>/dev/null
  tail -1 gpurun_out/r05_bench_$arch.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$arch', d['value'], d['ms_per_step'], d['config']['median_ms_200'], d['roofline']['frac'], d['roofline']['frac_is'])"
done
python3 bench.py --arch c5 > gpurun_out/r05_bench_c5.json 2>/dev/null
python3 bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
tail -1 gpurun_out/r05_bench_default.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['median_ms_200'], d['roofline']['frac'], d['roofline']['frac_is'], d['cpu_baseline']['value'], d['config']['tolerances']['measured'] is not None)"
du -sh gpurun_out
} > gpurun_out/r05_pipeline_final.txt 2>&1
tail -12 gpurun_out/r05_pipeline_final.txt
