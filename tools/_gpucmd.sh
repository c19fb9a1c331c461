python3 tools/step_timeline.py c2 64 20 2>&1 | grep -v amdgpu > gpurun_out/r04_step_timeline_c2.txt; tail -3 gpurun_out/r04_step_timeline_c2.txt
timeout 3000 python -m pytest tests -q -p no:cacheprovider -m gpu -x > gpurun_out/t_gpu.log 2>&1; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/t_gpu.log; grep -E "^E  " gpurun_out/t_gpu.log | head -8
cp gpurun_out/t_gpu.log gpurun_out/r04_gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python3 tools/dp2_one_gpu.py > gpurun_out/dp2.log 2>&1; tail -2 gpurun_out/dp2.log
