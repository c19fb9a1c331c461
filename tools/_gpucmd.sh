timeout 1500 python -m pytest tests/test_modules_gpu.py -q -p no:cacheprovider -k "ref144m4" > gpurun_out/t_mr4.log 2>&1; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/t_mr4.log; grep -E "^E  " gpurun_out/t_mr4.log | head -20
timeout 300 python bench.py --arch ref144 --no-cpu-baseline --no-kernels --no-variants 2>/dev/null | tail -1 | cut -c1-200
