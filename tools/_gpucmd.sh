for i in 1 2 3; do for cfg in "VPF_PREPROC_ON_SIDE=1" "VPF_PREPROC_ON_SIDE=0" "VPF_PREPROC_ON_SIDE=1 VPF_KV_BWD_ON_SIDE=0"; do
ms=$(env $cfg python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernels --no-variants 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
echo "$cfg  $ms ms/step"
done; done
for a in c3 c4 ref144; do for cfg in "VPF_PREPROC_ON_SIDE=1" "VPF_PREPROC_ON_SIDE=0"; do
ms=$(env $cfg python3 bench.py --arch $a --steps 40 --warmup 10 --no-cpu-baseline --no-kernels --no-variants 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
echo "$a $cfg  $ms ms/step"
done; done
