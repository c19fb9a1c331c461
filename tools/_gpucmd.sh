timeout 900 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "attention" > gpurun_out/t_attn.log 2>&1; grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/t_attn.log; grep -E "^E  " gpurun_out/t_attn.log | head -8
for lib in tools/_bin/lib_prev.so vipformer_amd/libvipformer_hip.so; do
VPF_LIB=$PWD/$lib python3 - <<'PY'
import torch, bench, os
from vipformer_amd import _lib as L, ops
H16=torch.float16
st = ops.rng.state("cuda")
out=[]
for tag,(B,Lq,Lkv,H) in (("c2",(128,96,1024,4)),("c3",(64,128,1024,4)),("ref144",(128,96,2048,4))):
    D=64*H
    q=torch.randn(B*Lq,D,device="cuda").to(H16); k=torch.randn(B*Lkv,D,device="cuda").to(H16); v=torch.randn(B*Lkv,D,device="cuda").to(H16)
    o=torch.empty_like(q); lse=torch.empty(B*H*Lq,device="cuda")
    L.call("vpf_attention_fwd", q, D, k, D, v, D, B, H, Lq, Lkv, 64, 0.125, 0.1, st, 7, o, D, lse)
    do=torch.randn(B*Lq,D,device="cuda").to(H16); dq=torch.empty_like(q); dk=torch.empty_like(k); dv=torch.empty_like(v); dl=torch.empty(B*H*Lq,device="cuda")
    us=bench._events(lambda: L.call("vpf_attention_bwd", q, D, k, D, v, D, o, D, do, D, lse, B, H, Lq, Lkv, 64, 0.125, 0.1, st, 7, dq, D, dk, D, dv, D, dl), 20, 3)
    out.append(f"{tag} {us:.1f}")
print(os.environ["VPF_LIB"].split("/")[-1], "CA bwd us:", " | ".join(out))
PY
done
bash tools/ab.sh "VPF_LIB=$PWD/tools/_bin/lib_prev.so" "VPF_LIB=$PWD/vipformer_amd/libvipformer_hip.so" 3 --steps 60
