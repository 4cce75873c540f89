"""Fixed cost vs per-round cost of the MFMA layer kernel: time at M = 1, 2, 4, 8 rounds of resident workgroups (768 per round at 3 per CU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd import _lib
lib = _lib.load(); dev = "cuda:0"
def bench(fn, n=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for K, N in [(256, 256), (256, 128), (128, 128)]:
    per_round_rows = 768 * 128 // (N // 128)
    row = []
    for rounds in (0.5, 1, 2, 3, 4, 8):
        M = int(per_round_rows * rounds)
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.06; b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
        t = bench(lambda: lib.bg_mlp_layer_forward(M, K, N, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), 1, _lib.current_stream_ptr()))
        row.append((rounds, M, round(t, 1), round(2.0 * M * K * N / t / 1e6, 1)))
    print(f"K={K} N={N}:", row, flush=True)
