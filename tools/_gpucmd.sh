set -x
timeout 1500 python -m pytest tests/test_boundary_gpu.py tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_trajectory_gpu.py "tests/test_modules_gpu.py::test_group2emb_first_conv_backward_fused_matches_two_kernels" "tests/test_modules_gpu.py::test_reference_script_geometry_takes_the_fused_paths" "tests/test_modules_gpu.py::test_training_step_with_dropout_vs_oracle" "tests/test_modules_gpu.py::test_models_vs_reference_golden" "tests/test_modules_gpu.py::test_stages_vs_reference_golden" -q -p no:cacheprovider > gpurun_out/gpu_tests4.log 2>&1; echo rc=$? >> gpurun_out/gpu_tests4.log
grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/gpu_tests4.log
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 300 gpurun_out/bench_default.err
timeout 300 python bench.py --arch ref144 --no-cpu-baseline --no-kernels --no-variants > gpurun_out/bench_ref144.json 2> gpurun_out/bench_ref144.err; tail -c 300 gpurun_out/bench_ref144.err
timeout 300 python bench.py --arch c5 --steps 20 --warmup 5 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; tail -c 300 gpurun_out/bench_c5.err
