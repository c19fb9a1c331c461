python3 tools/bench_head.py 2>&1 | grep -v "amdgpu.ids"
