"""A/B of two builds of the fp32-MFMA layer kernels in one process, alternating: the product library against tools/probe/libmlp_old.bin
(the previous epilogue: 4 x 4 quad transposes by DPP instead of the transposition through LDS)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd import _lib
lib = _lib.load(); dev = "cuda:0"
old = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libmlp_old.bin"))
old.bg_mlp_layer_forward.restype = C.c_int32
old.bg_mlp_layer_forward.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
old.bg_mlp_layer_backward.restype = C.c_int32
old.bg_mlp_layer_backward.argtypes = [C.c_int32] * 3 + [C.c_void_p] * 7
def bench(fn, n=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 98304
st = _lib.current_stream_ptr()
for K, N in [(256, 256), (256, 128), (128, 128), (64, 256)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.06; b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
    args = (M, K, N, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), 1, st)
    a, o = [], []
    for _ in range(3):
        a.append(bench(lambda: lib.bg_mlp_layer_forward(*args))); o.append(bench(lambda: old.bg_mlp_layer_forward(*args)))
    print(f"forward K={K} N={N}: new {min(a):6.1f} us, old {min(o):6.1f} us")
for K, N in [(256, 256), (128, 256), (128, 128)]:
    G = torch.randn(M, K, device=dev); Wt = torch.randn(N, K, device=dev) * 0.06; act = torch.nn.functional.elu(torch.randn(M, N, device=dev))
    out = torch.empty(M, N, device=dev); bg = torch.empty(N, device=dev); sc = torch.empty((M + 127) // 128 * N, device=dev)
    args = (M, K, N, _lib.ptr(G), _lib.ptr(Wt), _lib.ptr(act), _lib.ptr(out), _lib.ptr(bg), _lib.ptr(sc), st)
    a, o = [], []
    for _ in range(3):
        a.append(bench(lambda: lib.bg_mlp_layer_backward(*args))); o.append(bench(lambda: old.bg_mlp_layer_backward(*args)))
    print(f"backward K={K} N={N}: new {min(a):6.1f} us, old {min(o):6.1f} us")
