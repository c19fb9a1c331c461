python3 - <<'PY'
import torch, bench
torch.backends.cuda.matmul.allow_fp16_reduced_precision_reduction = False
M=12288
for (N,K) in ((768,256),(256,256),(512,256),(256,512),(512,131072//256)):
    for dt in (torch.float16,):
        dy=torch.randn(M,N,device="cuda",dtype=dt); x=torch.randn(M,K,device="cuda",dtype=dt)
        out=torch.empty(N,K,device="cuda",dtype=torch.float32)
        f=lambda: torch.mm(dy.t(), x)
        us=bench._events(f, 30, 5)
        print(f"torch mm dW[{N}x{K}] over M={M}: {us:.1f} us  {2*M*N*K/us/1e6:.0f} TFLOP/s", flush=True)
# the K/V wgrad: M = 131072, N = 512, K = 256
M=131072
dy=torch.randn(M,512,device="cuda",dtype=torch.float16); x=torch.randn(M,256,device="cuda",dtype=torch.float16)
us=bench._events(lambda: torch.mm(dy.t(), x), 20, 3)
print(f"torch mm dW[512x256] over M={M}: {us:.1f} us  {2*M*512*256/us/1e6:.0f} TFLOP/s")
# batched: 6 layers x 4 GEMMs as one bmm-like loop (sum of times)
M=12288
tot=0
for (N,K) in ((768,256),(256,256),(512,256),(256,512)):
    dy=torch.randn(M,N,device="cuda",dtype=torch.float16); x=torch.randn(M,K,device="cuda",dtype=torch.float16)
    tot+=bench._events(lambda: torch.mm(dy.t(), x), 30, 5)
print(f"one layer's four wgrads via torch.mm: {tot:.1f} us (ours grouped: ~41 us/layer, 210 us/7-layer stack)")
PY
