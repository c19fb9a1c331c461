#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export VPF_SCRATCH=/tmp/vpf_prof; mkdir -p $VPF_SCRATCH
for a in c3 c4 ref144 ref144m4; do
  bash tools/collect_step_bytes.sh r06 $a 2>&1 | tail -1
  cp gpurun_out/r06_step_bytes_$a.json profiles/
  python3 bench.py --arch $a --no-cpu-baseline > gpurun_out/r06_bench_$a.json 2> gpurun_out/r06_bench_$a.err
  tail -c 300 gpurun_out/r06_bench_$a.json | head -c 0; python3 -c "import json;d=json.loads(open('gpurun_out/r06_bench_$a.json').read().strip().splitlines()[-1]);print('$a',d['value'],d['ms_per_step'],d['roofline']['kernel'][:40],d['roofline']['frac'])"
done
python3 bench.py --arch c5 > gpurun_out/r06_bench_c5.json 2> gpurun_out/r06_bench_c5.err
python3 -c "import json;d=json.loads(open('gpurun_out/r06_bench_c5.json').read().strip().splitlines()[-1]);print('c5',d['value'],d['ms_per_step'])"
python3 bench.py --no-overlap --no-cpu-baseline --no-kernels --no-variants 2>/dev/null | tail -1 | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('single stream',d['ms_per_step'])"
python3 bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
python3 -c "import json;d=json.loads(open('gpurun_out/r06_bench_default.json').read().strip().splitlines()[-1]);print('c2',d['value'],d['ms_per_step'],d['roofline']['kernel'][:40],d['roofline']['frac'], d['roofline'].get('traffic'));print({k:(v['us_per_launch'],v['hbm_frac']) for k,v in d['kernels'].items()})"
