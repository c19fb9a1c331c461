timeout 3000 python -m pytest tests -q -p no:cacheprovider -m gpu -x > gpurun_out/t_gpu.log 2>&1; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/t_gpu.log; grep -E "^E  " gpurun_out/t_gpu.log | head -8
cp gpurun_out/t_gpu.log gpurun_out/r04_gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python3 tools/dp2_one_gpu.py > gpurun_out/dp2.log 2>&1; tail -1 gpurun_out/dp2.log
for a in c2 c3 c4 ref144 ref144m4; do bash tools/collect_step_bytes.sh r04 $a > gpurun_out/sb_$a.log 2>&1; tail -1 gpurun_out/sb_$a.log | cut -c1-150; rm -rf gpurun_out/stepbytes_r04_$a; done
cp gpurun_out/r04_step_bytes*.json profiles/
bash tools/collect_step_issue.sh r04 c2 > gpurun_out/si_c2.log 2>&1; tail -2 gpurun_out/si_c2.log | cut -c1-150; rm -rf gpurun_out/stepissue_r04_c2
bash tools/collect_profiles.sh r04 > gpurun_out/cp.log 2>&1; tail -2 gpurun_out/cp.log | cut -c1-150; rm -rf gpurun_out/prof_r04
cp gpurun_out/r04_pmc_summary.json profiles/
python3 bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/r04_bench_default.json | cut -c1-300
for a in ref144 c5 c3 c4; do python3 bench.py --arch $a --no-cpu-baseline > gpurun_out/r04_bench_$a.json 2>/dev/null; tail -1 gpurun_out/r04_bench_$a.json | cut -c1-200; done
python3 tools/step_timeline.py c2 64 20 2>&1 | grep -v amdgpu > gpurun_out/r04_step_timeline_c2.txt; tail -24 gpurun_out/r04_step_timeline_c2.txt
