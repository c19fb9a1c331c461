timeout 3000 python -m pytest tests -q -p no:cacheprovider -m gpu -x > gpurun_out/t_gpu.log 2>&1; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/t_gpu.log; grep -E "^E  " gpurun_out/t_gpu.log | head -8 | cut -c1-300
cp gpurun_out/t_gpu.log gpurun_out/r04_gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python3 tools/dp2_one_gpu.py > gpurun_out/dp2.log 2>&1; tail -1 gpurun_out/dp2.log
python3 bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/r04_bench_default.json | cut -c1-200
python3 bench.py --arch c5 --no-cpu-baseline > gpurun_out/r04_bench_c5.json 2>/dev/null; tail -1 gpurun_out/r04_bench_c5.json | cut -c1-120
