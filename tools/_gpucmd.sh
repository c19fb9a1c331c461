timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -q -p no:cacheprovider -x -k "g2e or group2emb or Group2Emb or training_step" > gpurun_out/t_g2e.log 2>&1; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/t_g2e.log; grep -E "^E  " gpurun_out/t_g2e.log | head -8
python3 - <<'PY'
import torch, bench, os
for lib in ("tools/_bin/lib_prev.so","vipformer_amd/libvipformer_hip.so"):
    pass
PY
for lib in tools/_bin/lib_prev.so vipformer_amd/libvipformer_hip.so; do echo $lib; VPF_LIB=$PWD/$lib python3 tools/microbench.py g2e 2>&1 | grep -v amdgpu | tail -3 | head -1; done
bash tools/ab.sh "VPF_LIB=$PWD/tools/_bin/lib_prev.so" "VPF_LIB=$PWD/vipformer_amd/libvipformer_hip.so" 3 --steps 60
