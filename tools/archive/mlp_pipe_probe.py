"""The persistent pipelined layer kernel (bg_mlp_pipe.hip) against the current one (bg_mlp.hip) at the training shapes: bit-equality of the outputs
and time alone on the GPU (HIP events, best of 5 x 20 launches), for several resident-workgroup counts."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess
import torch
from booster_gym_amd import _lib
lib = _lib.load(); dev = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "probe", "libmlp_pipe.bin")
if not os.path.isfile(so):  # build on the CPU side before gpurun (hipcc cross-compiles); the .bin travels with the snapshot
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-shared", "-o", so,
                           os.path.join(HERE, "probe", "mlp_pipe.hip")])
pipe = C.CDLL(so)
pipe.bg_mlp_layer_forward_pipe.restype = C.c_int32
pipe.bg_mlp_layer_forward_pipe.argtypes = [C.c_int32] * 3 + [C.c_void_p] * 4 + [C.c_int32, C.c_int32, C.c_void_p]
def bench(fn, n=20, reps=5):
    for _ in range(3): fn()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
M = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
torch.manual_seed(0)
for K, N in [(256, 256), (256, 128), (128, 128), (64, 256), (128, 256)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.06; b = torch.randn(N, device=dev)
    y0, y1 = torch.empty(M, N, device=dev), torch.full((M, N), float("nan"), device=dev)
    st = _lib.current_stream_ptr()
    _lib.check(lib.bg_mlp_layer_forward(M, K, N, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y0), 1, st))
    _lib.check(pipe.bg_mlp_layer_forward_pipe(M, K, N, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y1), 1, 0, st))
    torch.cuda.synchronize()
    same = bool(torch.equal(y0, y1))
    fl = 2.0 * M * K * N
    t0 = bench(lambda: lib.bg_mlp_layer_forward(M, K, N, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y0), 1, st))
    row = f"K={K:3d} N={N:3d}: bit-equal {same}  current {t0:6.1f} us ({fl/t0/1e6:5.1f} TF/s) | pipelined"
    for wg in (256, 384, 512, 768):
        t1 = bench(lambda: pipe.bg_mlp_layer_forward_pipe(M, K, N, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y1), 1, wg, st))
        row += f"  {wg} wg: {t1:6.1f} us ({fl/t1/1e6:5.1f})"
    print(row, flush=True)
    if not same:
        print("   max abs diff", float((y0 - y1).abs().max()), "nan", int(torch.isnan(y1).sum()))
