#!/bin/bash
# round 5 profile pipeline, part 1: whole-step byte budget + issue counters + stand-alone PMC summary + timeline + the bench line (c2)
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
python3 tools/step_timeline.py > gpurun_out/r05_step_timeline_c2.txt 2>/dev/null; tail -30 gpurun_out/r05_step_timeline_c2.txt
python3 bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
tail -1 gpurun_out/r05_bench_default.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['median_ms_200'], d['roofline']['frac'], d['roofline']['frac_is'], d['cpu_baseline']['value'])"
du -sh gpurun_out
} > gpurun_out/r05_pipeline1.txt 2>&1
tail -40 gpurun_out/r05_pipeline1.txt
