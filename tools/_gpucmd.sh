python __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python -m pytest tests/test_boundary_gpu.py -q -p no:cacheprovider -k "gradscaler or loss_scale" > gpurun_out/t_gs.log 2>&1; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/t_gs.log; grep -E "^E  " gpurun_out/t_gs.log | head -12
