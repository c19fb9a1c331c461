python3 tools/diag_ft_scale.py 16 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider\|print(f"
timeout 1500 python -m pytest tests/test_partseg_gpu.py tests/test_modules_gpu.py -q -p no:cacheprovider -k "partseg or finetune or ft_" > gpurun_out/t_ft.log 2>&1; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/t_ft.log; grep -E "^E  " gpurun_out/t_ft.log | head -12
