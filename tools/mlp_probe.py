"""Fused fp32-MFMA layer kernels against torch (hipBLASLt) at the training shapes: forward Linear+bias+ELU and backward Linear^T+ELU'+bias-grad, HIP events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd import _lib
lib = _lib.load(); dev = "cuda:0"
def bench(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 98304
for K, N in [(256, 256), (256, 128), (128, 128), (64, 256)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.06; b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
    fl = 2.0 * M * K * N
    t_lib = bench(lambda: torch.nn.functional.elu_(torch.addmm(b, x, w.t(), out=y)))
    t_gemm = bench(lambda: torch.addmm(b, x, w.t(), out=y))
    t_mine = bench(lambda: lib.bg_mlp_layer_forward(M, K, N, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), 1, _lib.current_stream_ptr()))
    print(f"K={K} N={N}: torch addmm {t_gemm:7.1f} us ({fl/t_gemm/1e6:5.1f} TF/s)  addmm+elu_ {t_lib:7.1f} us   fused MFMA kernel {t_mine:7.1f} us ({fl/t_mine/1e6:5.1f} TF/s)")
print("backward: Gout = (G W) * elu'(a), + column sums")
for K, N in [(256, 256), (128, 256), (128, 128)]:
    G = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev) * 0.06; a = torch.nn.functional.elu(torch.randn(M, N, device=dev))
    Wt = W.t().contiguous(); out = torch.empty(M, N, device=dev); bgd = torch.zeros(N, device=dev); scr = torch.empty(((M + 127) // 128) * N, device=dev)
    def lib_path():
        torch.mm(G, W, out=out)
        lib.bg_elu_backward_colsum(M, N, _lib.ptr(out), _lib.ptr(a), _lib.ptr(bgd), _lib.ptr(scr), _lib.current_stream_ptr())
    t_lib = bench(lib_path)
    t_mm = bench(lambda: torch.mm(G, W, out=out))
    t_mine = bench(lambda: lib.bg_mlp_layer_backward(M, K, N, _lib.ptr(G), _lib.ptr(Wt), _lib.ptr(a), _lib.ptr(out), _lib.ptr(bgd), _lib.ptr(scr), _lib.current_stream_ptr()))
    fl = 2.0 * M * K * N
    print(f"K={K} N={N}: torch mm {t_mm:7.1f} us  mm+elu_bwd_colsum {t_lib:7.1f} us   fused bwd kernel {t_mine:7.1f} us ({fl/t_mine/1e6:5.1f} TF/s)")
