run() { env $1 python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernels --no-variants 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['median_ms_200']['median'])"; }
for i in 1 2 3; do
  for cfg in "VPF_X=0" "VPF_LIB=tools/_bin/libvipformer_abl1.so" "VPF_LIB=tools/_bin/libvipformer_abl2.so" "VPF_LIB=tools/_bin/libvipformer_abl3.so"; do
    echo "$cfg  $(run "$cfg")"
  done
done
