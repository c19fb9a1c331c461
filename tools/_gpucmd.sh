timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -q -p no:cacheprovider -x > gpurun_out/t_all.log 2>&1; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/t_all.log; grep -E "^E  " gpurun_out/t_all.log | head -8
bash tools/ab.sh "VPF_LIB=$PWD/tools/_bin/lib_head.so" "VPF_LIB=$PWD/vipformer_amd/libvipformer_hip.so" 3 --steps 60
