"""Stand-alone timing of the weight-gradient kernel (bg_mlp_weight_grad) at the six layer shapes of the update, against the library path it
replaces (split-K torch.bmm + torch.sum), on an otherwise idle GPU.  python tools/wgrad_probe.py [M]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd import _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
lib = _lib.load()
dev = "cuda:0"


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out = {}
for C_out, C_in, C_real in [(256, 64, 47), (128, 256, 256), (128, 128, 128), (256, 64, 61), (256, 256, 256)]:
    G = torch.randn(M, C_out, device=dev); A = torch.randn(M, C_in, device=dev); A[:, C_real:] = 0
    dW = torch.empty(C_out, C_real, device=dev)
    ntiles = (C_out // 128) * max(1, C_in // 128)
    res = {}
    for wgs in (128, 256, 512):
        s = max(8, wgs // ntiles // 8 * 8)
        sc = torch.empty(s * C_out * C_in, device=dev)
        us = timeit(lambda: _lib.check(lib.bg_mlp_weight_grad(M, C_out, C_in, C_real, _lib.ptr(G), _lib.ptr(A), _lib.ptr(dW), _lib.ptr(sc), s, _lib.current_stream_ptr())))
        res[f"hip_{wgs}wg_us"] = round(us, 1)
    S = 32
    dws = torch.empty(S, C_out, C_in, device=dev); dsum = torch.empty(C_out, C_in, device=dev)
    def libpath():
        torch.bmm(G.view(S, M // S, C_out).transpose(1, 2), A.view(S, M // S, C_in), out=dws)
        torch.sum(dws, dim=0, out=dsum)
    res["hipblaslt_bmm_sum_us"] = round(timeit(libpath), 1)
    flop = 2.0 * M * C_out * C_in
    best = min(v for k, v in res.items() if k.startswith("hip_"))
    res["best_tflops"] = round(flop / best / 1e6, 1); res["frac_of_157.3"] = round(flop / best / 1e6 / 157.3, 3)
    out[f"{C_out}x{C_in}"] = res
    print(f"{C_out}x{C_in}: {res}", flush=True)
json.dump({"M": M, "shapes": out}, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wgrad_probe.json"), "w"), indent=1)
