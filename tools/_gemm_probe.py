import torch, time, os
dev="cuda:0"
B=98304
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
print("preferred blas:", torch.backends.cuda.preferred_blas_library())
for (K_in, out) in [(256,256),(61,256),(256,128),(128,128),(47,256),(128,12),(128,1)]:
    X=torch.randn(B,K_in,device=dev); dY=torch.randn(B,out,device=dev); W=torch.randn(out,K_in,device=dev); b=torch.randn(out,device=dev)
    fl=2*B*K_in*out
    t=bench(lambda: torch.mm(dY.t(), X)); print(f"dW {K_in}x{out}: mm(dY^T,X) {t:8.1f} us {fl/t/1e6:6.1f} TF/s")
    for S in (16,32,64,128,256):
        Xs=X.view(S,B//S,K_in); dYs=dY.view(S,B//S,out)
        t=bench(lambda: torch.bmm(dYs.transpose(1,2), Xs).sum(0)); print(f"     bmm split S={S:4d} {t:8.1f} us {fl/t/1e6:6.1f} TF/s")
    t=bench(lambda: torch.addmm(b, X, W.t())); print(f"fwd addmm           {t:8.1f} us {fl/t/1e6:6.1f} TF/s")
    t=bench(lambda: torch.mm(dY, W)); print(f"dX  mm(dY,W)        {t:8.1f} us {fl/t/1e6:6.1f} TF/s")
    t=bench(lambda: dY.sum(0)); print(f"db sum              {t:8.1f} us")
H=torch.randn(B,256,device=dev)
t=bench(lambda: torch.nn.functional.elu(H)); print("elu fwd 256", t, "us", B*256*8/t/1e3, "GB/s")
