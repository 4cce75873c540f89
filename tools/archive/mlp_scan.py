"""Launch time of the fused forward layer (256 x 256) over M, in units of dispatch rounds: where the per-round prologue / epilogue overhead shows."""
import os, sys
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
K, N = 256, 256
for rounds in (0.25, 0.5, 1, 1.5, 2, 3, 4, 8):
    M = int(rounds * 768 * 128 / 2)   # 768 resident WGs, N=256 -> 2 WGs per slab
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.06; b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
    t = bench(lambda: lib.bg_mlp_layer_forward(M, K, N, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), 1, _lib.current_stream_ptr()))
    print(f"rounds {rounds:5.2f} M={M:7d}: {t:7.1f} us  {2.0*M*K*N/t/1e6:6.1f} TF/s")
