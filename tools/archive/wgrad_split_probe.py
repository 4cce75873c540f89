"""The grouped weight-gradient launch alone on the GPU: fp32-MFMA kernel (rows per wave, and rows shared per workgroup) against the split-bf16 form
(9 and 6 products, rows shared through LDS), at the update's six layers, M = 98,304."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd import _lib
from booster_gym_amd.utils.model import plan_wgrad_slices
lib = _lib.load(); dev = "cuda:0"; st = _lib.current_stream_ptr()
M = 98304
shapes = [(256, 64, 61), (256, 256, 256), (128, 256, 256), (256, 64, 47), (128, 256, 256), (128, 128, 128)]
def build(share):
    slices, tw = plan_wgrad_slices([(co, ci) for co, ci, _ in shapes], M, 256, share_rows=share)
    arr = (_lib.WgradProblem * len(shapes))(); keep = []
    for k, ((co, ci, cr), sl) in enumerate(zip(shapes, slices)):
        G = torch.randn(M, co, device=dev); A = torch.randn(M, ci, device=dev); A[:, cr:] = 0
        dW = torch.empty(co, cr, device=dev); sc = torch.empty(sl * co * ci, device=dev); keep.append((G, A, dW, sc))
        arr[k].G, arr[k].A, arr[k].dW, arr[k].scratch = G.data_ptr(), A.data_ptr(), dW.data_ptr(), sc.data_ptr()
        arr[k].M, arr[k].C_out, arr[k].C_in, arr[k].C_in_real, arr[k].slices, arr[k].tiles_per_workgroup = M, co, ci, cr, sl, tw[k]
    return arr, keep, slices, tw
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)
flop = sum(2.0 * M * co * ci for co, ci, _ in shapes)
a0, k0, s0, t0 = build(False)
a1, k1, s1, t1 = build(True)
out = {"gflop": round(flop / 1e9, 1), "slices_rows_per_wave": s0, "slices_rows_shared": s1}
out["fp32_rows_per_wave_us"] = bench(lambda: lib.bg_mlp_weight_grad_group(a0, 6, st))
out["fp32_rows_shared_us"] = bench(lambda: lib.bg_mlp_weight_grad_group(a1, 6, st))
for terms in (9, 6):
    rc = lib.bg_mlp_weight_grad_group_split(a1, 6, terms, st)
    assert rc == 0, lib.bg_last_error()
    out[f"split{terms}_us"] = bench(lambda: lib.bg_mlp_weight_grad_group_split(a1, 6, terms, st))
print(json.dumps(out))
