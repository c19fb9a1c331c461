timeout 1500 python -m pytest tests/test_modules_gpu.py tests/test_boundary_gpu.py -q -p no:cacheprovider -x > gpurun_out/t_mod.log 2>&1; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/t_mod.log; grep -E "^E  " gpurun_out/t_mod.log | head -8
bash tools/ab.sh "VPF_WGRAD_CARRY=0" "VPF_WGRAD_CARRY=1" 4 --steps 60
