for cfg in "VPF_PREPROC_ON_SIDE=0" "VPF_PREPROC_ON_SIDE=1"; do echo "=== $cfg"; env $cfg python3 tools/step_timeline.py c2 64 20 2>&1 | grep -v amdgpu | tail -24; done
