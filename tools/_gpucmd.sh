timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -q -p no:cacheprovider -x -k "sa_ or fused_sa or stack or encoder or training_step" > gpurun_out/t_sa.log 2>&1; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/t_sa.log; grep -E "^E  " gpurun_out/t_sa.log | head -8
for lib in tools/_bin/lib_prev.so vipformer_amd/libvipformer_hip.so; do
VPF_LIB=$PWD/$lib python3 - <<'PY'
import torch, bench, os
legs=bench.kernel_legs(torch.device("cuda",0), bench.ARCHS["c2"], 64)
print(os.environ["VPF_LIB"].split("/")[-1], {k:round(v["us_per_launch"],2) for k,v in legs.items() if "sa_" in k})
PY
done
bash tools/ab.sh "VPF_LIB=$PWD/tools/_bin/lib_prev.so" "VPF_LIB=$PWD/vipformer_amd/libvipformer_hip.so" 3 --steps 60
