timeout 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 200 gpurun_out/bench_default.err
for a in c3 c4; do timeout 300 python bench.py --arch $a --no-cpu-baseline --no-variants > gpurun_out/bench_$a.json 2>/dev/null; done
