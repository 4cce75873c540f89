"""Per-layer time of the split weight-gradient launch: each of the six layers ALONE with the slices the six-layer plan gives it (its workgroups then run
on an otherwise idle chip: a lower bound of what each layer's workgroups need inside the grouped launch).   gpurun -- python tools/wgrad_split_layers_probe.py"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd import _lib
from booster_gym_amd.utils.model import plan_wgrad_slices
lib = _lib.load(); dev = "cuda:0"; st = _lib.current_stream_ptr()
M = 98304
six = [(128, 256, 256), (256, 256, 256), (256, 64, 61), (128, 128, 128), (128, 256, 256), (256, 64, 47)]
slices, tw = plan_wgrad_slices([(co, ci) for co, ci, _ in six], M, 256, share_rows=True)
g = torch.Generator(device="cpu").manual_seed(3)
def bench(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)
for k, ((co, ci, cr), sl) in enumerate(zip(six, slices)):
    G = (torch.randn(M, co, generator=g) * 0.01).to(dev)
    A = torch.zeros(M, ci); A[:, :cr] = torch.nn.functional.elu(torch.randn(M, cr, generator=g)); A = A.to(dev)
    dW = torch.empty(co, cr, device=dev); sc = torch.empty(sl * co * ci, device=dev)
    arr = (_lib.WgradProblem * 1)()
    arr[0].G, arr[0].A, arr[0].dW, arr[0].scratch = G.data_ptr(), A.data_ptr(), dW.data_ptr(), sc.data_ptr()
    arr[0].M, arr[0].C_out, arr[0].C_in, arr[0].C_in_real, arr[0].slices, arr[0].tiles_per_workgroup = M, co, ci, cr, sl, tw[k]
    us = bench(lambda: _lib.check(lib.bg_mlp_weight_grad_group_split_partial(arr, 1, 9, st), "split"))
    rows_per_wave = M / sl / (4 // tw[k])
    print(json.dumps({"layer": f"{co}x{ci}", "slices": sl, "tiles_per_workgroup": tw[k], "rows_per_wave": round(rows_per_wave), "us_alone": us}), flush=True)
