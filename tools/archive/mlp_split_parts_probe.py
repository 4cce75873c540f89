"""Where does the split-bf16 layer kernel's time go?  Probe builds of bg_mlp_split.hip (tools/probe/libsplit_terms*.bin: -DBG_PROBE_TERMS allows
terms = 1 = only the hi x hi product, 1/9 of the MFMAs with everything else in place; -DBG_PROBE_NO_STORE predicates the epilogue stores off)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd import _lib
lib = _lib.load(); dev = "cuda:0"
here = os.path.dirname(os.path.abspath(__file__))
def load(name):
    l = C.CDLL(os.path.join(here, "probe", name))
    l.bg_mlp_layer_forward_split.restype = C.c_int32
    l.bg_mlp_layer_forward_split.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    return l
pt, pn = load("libsplit_terms.bin"), load("libsplit_terms_nostore.bin")
def bench(fn, n=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)
M = 98304
for K, N in [(256, 256), (256, 128), (128, 128), (64, 256)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.06; b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
    planes = torch.empty(N * K * 3, dtype=torch.int16, device=dev)
    st = _lib.current_stream_ptr()
    _lib.check(lib.bg_mlp_split_weights(N, K, _lib.ptr(w), K, N, K, 0, _lib.ptr(planes), st), "split")
    a = lambda t: (M, K, N, _lib.ptr(x), _lib.ptr(planes), _lib.ptr(b), _lib.ptr(y), 1, t, st)
    out = {"shape": f"K={K} N={N}"}
    for t in (9, 6, 1):
        out[f"terms{t}_us"] = bench(lambda: pt.bg_mlp_layer_forward_split(*a(t)))
        out[f"terms{t}_nostore_us"] = bench(lambda: pn.bg_mlp_layer_forward_split(*a(t)))
    out["mfma_us_per_term_at_2.2GHz"] = round(2.0 * M * K * N / 2.5e15 * 2.4 / 2.2 * 1e6, 1)
    print(json.dumps(out), flush=True)
