for i in 1 2 3; do for cfg in "VPF_PREPROC_ON_SIDE=1" "VPF_PREPROC_ON_SIDE=0" "VPF_PREPROC_ON_SIDE=0 VPF_MAIN_FIRST=1 VPF_KV_FWD_ON_SIDE=1"; do
ms=$(env $cfg python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernels --no-variants 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
echo "$cfg  $ms ms/step"
done; done
VPF_PREPROC_ON_SIDE=0 VPF_MAIN_FIRST=1 VPF_KV_FWD_ON_SIDE=1 python3 tools/step_timeline.py c2 64 30 2>&1 | grep -v amdgpu | tail -26
