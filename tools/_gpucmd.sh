bash tools/collect_step_bytes.sh r04 c2 > gpurun_out/sb_c2.log 2>&1; tail -5 gpurun_out/sb_c2.log | cut -c1-200
rm -rf gpurun_out/stepbytes_r04_c2
