"""How much of the fused layer kernel is its store tail?  The same kernel built with its epilogue stores disabled (tools/probe/libmlp_nostore.bin,
-DBG_PROBE_NO_STORE) against the product build, forward layers at the training shapes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd import _lib
lib = _lib.load(); dev = "cuda:0"
ns = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libmlp_nostore.bin"))
ns.bg_mlp_layer_forward.restype = C.c_int32
ns.bg_mlp_layer_forward.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
def bench(fn, n=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 98304
for K, N in [(256, 256), (256, 128), (128, 128), (64, 256)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.06; b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
    args = (M, K, N, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), 1, _lib.current_stream_ptr())
    t1 = bench(lambda: lib.bg_mlp_layer_forward(*args)); t0 = bench(lambda: ns.bg_mlp_layer_forward(*args))
    print(f"K={K} N={N}: product {t1:6.1f} us, without the stores {t0:6.1f} us, MFMA time at 2.4 GHz {2.0*M*K*N/157.3e6:6.1f} us")
