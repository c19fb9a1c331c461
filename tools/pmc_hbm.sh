#!/bin/bash
# HBM traffic per kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (TCC slot budget),
# then per-kernel averages.  usage: tools/pmc_hbm.sh <tag> <microbench targets...>
tag=$1; shift
export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=$PWD/gpurun_out/hbm_${tag}_$ctr
  mkdir -p $out
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o $tag -- python3 tools/microbench.py "$@" > $out/log.txt 2>&1
done
python3 - $PWD/gpurun_out/hbm_${tag}_FETCH_SIZE/${tag}_counter_collection.csv $PWD/gpurun_out/hbm_${tag}_WRITE_SIZE/${tag}_counter_collection.csv > $PWD/gpurun_out/${tag}_hbm_traffic.txt <<'PY'
import csv, sys, collections
def load(path):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        key = (r['Kernel_Name'], r['Grid_Size'] if 'Grid_Size' in r else '')
        agg.setdefault(key, []).append(float(r['Counter_Value']))
    return agg
f, w = load(sys.argv[1]), load(sys.argv[2])
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); units: KiB per dispatch (median).")
print("# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> 'fetch_x2' column.")
print("kernel | grid | n | FETCH_SIZE_KiB | fetch_x2_KiB | WRITE_SIZE_KiB")
for key, fv in f.items():
    wv = w.get(key, [0.0])
    med = lambda v: sorted(v)[len(v) // 2]
    if len(fv) >= 3:
        print(f"{key[0][:90]} | {key[1]} | {len(fv)} | {med(fv):.1f} | {2*med(fv):.1f} | {med(wv):.1f}")
PY
cat $PWD/gpurun_out/${tag}_hbm_traffic.txt | cut -c1-200
