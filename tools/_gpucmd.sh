for a in c3 c4; do for i in 1 2; do for m in 512 1; do
ms=$(VPF_ATTN_CA_MERGED=$m python3 bench.py --arch $a --steps 40 --warmup 10 --no-cpu-baseline --no-kernels --no-variants 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
echo "$a VPF_ATTN_CA_MERGED=$m  $ms ms/step"
done; done; done
