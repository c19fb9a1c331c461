bash tools/ab.sh "VPF_KV_BWD_ON_SIDE=0" "VPF_KV_BWD_ON_SIDE=1" 4 --steps 60
python3 tools/step_timeline.py c2 64 30 2>&1 | tail -12
