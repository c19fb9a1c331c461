for lib in tools/_bin/lib_prev.so vipformer_amd/libvipformer_hip.so; do
VPF_LIB=$PWD/$lib python3 - <<'PY'
import torch, bench, os
from vipformer_amd import ops, _lib as L
H16=torch.float16
out=[]
for (M,N,K) in ((64,256,512),(128,256,512),(128,512,512)):
    a=torch.randn(M,K,device="cuda").to(H16); w=torch.randn(N,K,device="cuda").to(H16)
    f=bench._events(lambda: ops.linear_fwd(a, w, N, K, None, out_f32=True), 50, 5)
    dy=torch.randn(M,N,device="cuda").to(H16)
    d=bench._events(lambda: ops.linear_dgrad(dy, w, N, K, out_f32=True), 50, 5)
    out.append(f"{M}x{N}x{K}: fwd {f:.2f} dgrad {d:.2f}")
B,Lt,D=128,96,256
x=torch.randn(B,Lt,D,device="cuda"); o=torch.empty(B,2*D,device="cuda"); arg=torch.empty(B,D,dtype=torch.int32,device="cuda"); dx=torch.empty_like(x); dout=torch.randn(B,2*D,device="cuda")
pf=bench._events(lambda: L.call("vpf_pool_fwd", x, B, Lt, D, o, arg), 50, 5); pb=bench._events(lambda: L.call("vpf_pool_bwd", dout, arg, B, Lt, D, dx), 50, 5)
print(os.environ["VPF_LIB"].split("/")[-1], " | ".join(out), f"| pool fwd {pf:.2f} bwd {pb:.2f}")
PY
done
