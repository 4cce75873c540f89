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
# all six layers of both networks in one grouped launch (what the training loop runs)
from booster_gym_amd.utils.model import plan_wgrad_slices
shapes = [(256, 64, 61), (256, 256, 256), (128, 256, 256), (256, 64, 47), (128, 256, 256), (128, 128, 128)]
for wgs, share in ((256, False), (256, True), (248, True), (512, True)):
    sl, tw = plan_wgrad_slices([(co, ci) for co, ci, _ in shapes], M, wgs, share_rows=share)
    arr = (_lib.WgradProblem * len(shapes))(); keep = []
    for k, ((co, ci, cr), s_) in enumerate(zip(shapes, sl)):
        G = torch.randn(M, co, device=dev); A = torch.randn(M, ci, device=dev); dW = torch.empty(co, cr, device=dev); sc = torch.empty(s_ * co * ci, device=dev)
        keep.append((G, A, dW, sc))
        arr[k].G, arr[k].A, arr[k].dW, arr[k].scratch = G.data_ptr(), A.data_ptr(), dW.data_ptr(), sc.data_ptr()
        arr[k].M, arr[k].C_out, arr[k].C_in, arr[k].C_in_real, arr[k].slices, arr[k].tiles_per_workgroup = M, co, ci, cr, s_, tw[k]
    us = timeit(lambda: _lib.check(lib.bg_mlp_weight_grad_group(arr, len(shapes), _lib.current_stream_ptr())))
    flop = sum(2.0 * M * co * ci for co, ci, _ in shapes)
    out[f"grouped_{wgs}wg_share{int(share)}"] = {"us": round(us, 1), "slices": sl, "tiles_per_workgroup": tw, "tflops": round(flop / us / 1e6, 1), "frac_of_157.3": round(flop / us / 1e6 / 157.3, 3)}
    print(f"grouped, {wgs} workgroups, share_rows={share}: {out[f'grouped_{wgs}wg_share{int(share)}']}", flush=True)
json.dump({"M": M, "shapes": out}, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wgrad_probe.json"), "w"), indent=1)
