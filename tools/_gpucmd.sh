timeout 900 python -m pytest tests/test_partseg_gpu.py -q -p no:cacheprovider -x -k "graphed" > gpurun_out/t_gs.log 2>&1; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/t_gs.log; grep -E "^E  " gpurun_out/t_gs.log | head -12
python3 bench.py --arch c5 --no-cpu-baseline > gpurun_out/c5.log 2>&1; tail -1 gpurun_out/c5.log | cut -c1-1200
