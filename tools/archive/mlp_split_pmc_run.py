"""Workload for tools/mlp_split_pmc.sh: the 256 x 256 forward layer, fp32-MFMA kernel and split kernels (9 and 6 terms), a few launches each."""
import sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from booster_gym_amd import _lib
lib = _lib.load(); dev = "cuda:0"; st = _lib.current_stream_ptr()
M, K, N = 98304, 256, 256
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.06; b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
planes = torch.empty(N * K * 3, dtype=torch.int16, device=dev)
_lib.check(lib.bg_mlp_split_weights(N, K, _lib.ptr(w), K, N, K, 0, _lib.ptr(planes), st), "split")
for _ in range(4):
    lib.bg_mlp_layer_forward(M, K, N, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), 1, st)
    lib.bg_mlp_layer_forward_split(M, K, N, _lib.ptr(x), _lib.ptr(planes), _lib.ptr(b), _lib.ptr(y), 1, 9, st)
    lib.bg_mlp_layer_forward_split(M, K, N, _lib.ptr(x), _lib.ptr(planes), _lib.ptr(b), _lib.ptr(y), 1, 6, st)
torch.cuda.synchronize()
