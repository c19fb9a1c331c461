for i in 1 2 3; do for cfg in "VPF_KV_GATE=1" "VPF_KV_GATE=0" "VPF_KV_GATE=2" "VPF_KV_BWD_ON_SIDE=0"; do
ms=$(env $cfg python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernels --no-variants 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
echo "$cfg  $ms ms/step"
done; done
