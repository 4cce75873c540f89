"""bg_mlp_chain_forward_group alone on the GPU: critic, actor, and both in one launch (critic slabs first), against the per-layer launches."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd import _lib
lib = _lib.load(); dev = "cuda:0"
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
torch.manual_seed(0)
def net(M, dims):
    K0, N1, N2, N3 = dims
    x = torch.randn(M, K0, device=dev)
    Ws = [torch.randn(n, k, device=dev) / k ** 0.5 for k, n in ((K0, N1), (N1, N2), (N2, N3))]
    bs = [torch.randn(n, device=dev) * 0.1 for n in (N1, N2, N3)]
    ys = [torch.empty((M + 127) // 128 * 128, n, device=dev) for n in (N1, N2, N3)]
    p = _lib.ptr
    d = _lib.MlpChain(M, K0, N1, N2, N3, 0, p(x), p(Ws[0]), p(bs[0]), p(Ws[1]), p(bs[1]), p(Ws[2]), p(bs[2]), p(ys[0]), p(ys[1]), p(ys[2]))
    return d, (x, Ws, bs, ys)
dc, kc = net(102400, (64, 256, 256, 128))
da, ka = net(98304, (64, 256, 128, 128))
st = _lib.current_stream_ptr()
def group(ds):
    arr = (_lib.MlpChain * len(ds))(*ds)
    return lambda: _lib.check(lib.bg_mlp_chain_forward_group(C.addressof(arr), len(ds), st))
def layers(k):
    x, Ws, bs, ys = k
    def f():
        hin = x
        for l in range(3):
            _lib.check(lib.bg_mlp_layer_forward(x.shape[0], hin.shape[1], Ws[l].shape[0], _lib.ptr(hin), _lib.ptr(Ws[l]), _lib.ptr(bs[l]), _lib.ptr(ys[l]), 1, st))
            hin = ys[l][: x.shape[0]]
    return f
fc, fa = layers(kc), layers(ka)
print(f"chain: critic {bench(group([dc])):6.1f} us  actor {bench(group([da])):6.1f} us  critic+actor in one launch {bench(group([dc, dc.__class__.from_buffer_copy(da)])):6.1f} us  "
      f"actor+critic {bench(group([da, dc])):6.1f} us | per-layer launches: critic {bench(fc):6.1f} us  actor {bench(fa):6.1f} us  both {bench(lambda: (fc(), fa())):6.1f} us", flush=True)
