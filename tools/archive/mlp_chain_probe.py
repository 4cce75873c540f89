"""The register-chained forward kernel of one network (tools/probe/mlp_chain.hip) against the three per-layer launches of bg_mlp.hip: outputs
against torch (fp64) and time alone on the GPU (HIP events, best of 5 x 20 launches)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess
import torch
from booster_gym_amd import _lib
lib = _lib.load(); dev = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "..", "probe", "libmlp_chain.bin")
if not os.path.isfile(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-mllvm",
                           "-amdgpu-sched-strategy=max-ilp", "-shared", "-o", so, os.path.join(HERE, "mlp_chain_probe_v1.hip")])
ch = C.CDLL(so)
ch.bg_mlp_chain_forward.restype = C.c_int32
ch.bg_mlp_chain_forward.argtypes = [C.c_int32] * 5 + [C.c_void_p] * 10 + [C.c_int32, C.c_void_p]
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
for name, M, dims in (("actor", 98304, (64, 256, 128, 128)), ("critic", 102400, (64, 256, 256, 128)), ("actor ragged", 98304 - 77, (64, 256, 128, 128))):
    K0, N1, N2, N3 = dims
    x = torch.randn(M, K0, device=dev)
    Ws = [torch.randn(n, k, device=dev) / k ** 0.5 for k, n in ((K0, N1), (N1, N2), (N2, N3))]
    bs = [torch.randn(n, device=dev) * 0.1 for n in (N1, N2, N3)]
    ys = [torch.full(((M + 127) // 128 * 128, n), float("nan"), device=dev) for n in (N1, N2, N3)]
    zs = [torch.empty(M, n, device=dev) for n in (N1, N2, N3)]
    st = _lib.current_stream_ptr()
    def chain(wg=256):
        rc = ch.bg_mlp_chain_forward(M, K0, N1, N2, N3, _lib.ptr(x), _lib.ptr(Ws[0]), _lib.ptr(bs[0]), _lib.ptr(Ws[1]), _lib.ptr(bs[1]), _lib.ptr(Ws[2]),
                                     _lib.ptr(bs[2]), _lib.ptr(ys[0]), _lib.ptr(ys[1]), _lib.ptr(ys[2]), wg, st)
        assert rc == 0, rc
    def layers():
        hin = x
        for l in range(3):
            _lib.check(lib.bg_mlp_layer_forward(M, hin.shape[1], Ws[l].shape[0], _lib.ptr(hin), _lib.ptr(Ws[l]), _lib.ptr(bs[l]), _lib.ptr(zs[l]), 1, st))
            hin = zs[l]
    chain(); layers(); torch.cuda.synchronize()
    sub = slice(0, 8192)
    ref = x[sub].double()
    errs = []
    for l in range(3):
        ref = torch.nn.functional.elu(ref @ Ws[l].double().t() + bs[l].double())
        errs.append((float((ys[l][sub].double() - ref).abs().max()), float((zs[l][sub].double() - ref).abs().max())))
    ys = [y[:M] for y in ys]; nan = sum(int(torch.isnan(y).sum()) for y in ys)
    dmax = max(float((y - z).abs().max()) for y, z in zip(ys, zs))
    fl = 2.0 * M * (K0 * N1 + N1 * N2 + N2 * N3)
    t0 = bench(layers); t1 = bench(chain); t2 = bench(lambda: chain(0)); t0b = bench(layers)
    print(f"{name:13s} M={M}: chain 256 wg {t1:6.1f} us ({fl/t1/1e6:5.1f} TF/s), one wg per slab {t2:6.1f} us | three launches {t0:6.1f} / {t0b:6.1f} us ({fl/t0/1e6:5.1f} TF/s) | nan {nan} | chain vs launches max abs {dmax:.2e}"
          f" | vs fp64 (chain, launches) per layer {['%.1e %.1e' % e for e in errs]}", flush=True)
