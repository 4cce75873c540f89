"""The opt-in split weight-gradient launch (bg_mlp_weight_grad_group_split, 9 products) beside the fp32-MFMA launch on loop-like operands (activations
= ELU outputs, gradients ~ N(0, 0.01)): launch times alone, and the error against float64 split into its parts -- rms, mean signed error, and the
correlation of the error with the sign of the result (an accumulator that truncates pulls every sum towards -infinity or towards zero; an unbiased one
does neither).   gpurun -- python tools/wgrad_split_error_probe.py"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd import _lib
from booster_gym_amd.utils.model import plan_wgrad_slices
lib = _lib.load(); dev = "cuda:0"; st = _lib.current_stream_ptr()
M = 98304
six = [(128, 256, 256), (256, 256, 256), (256, 64, 61), (128, 128, 128), (128, 256, 256), (256, 64, 47)]
def build(shapes, share):
    slices, tw = plan_wgrad_slices([(co, ci) for co, ci, _ in shapes], M, 256, share_rows=share)
    arr = (_lib.WgradProblem * len(shapes))(); keep = []
    g = torch.Generator(device="cpu").manual_seed(3)
    for k, ((co, ci, cr), sl) in enumerate(zip(shapes, slices)):
        G = (torch.randn(M, co, generator=g) * 0.01).to(dev)
        A = torch.zeros(M, ci); A[:, :cr] = torch.nn.functional.elu(torch.randn(M, cr, generator=g)); A = A.to(dev)
        dW = torch.empty(co, cr, device=dev); sc = torch.empty(sl * co * ci, device=dev); keep.append((G, A, dW, sc))
        arr[k].G, arr[k].A, arr[k].dW, arr[k].scratch = G.data_ptr(), A.data_ptr(), dW.data_ptr(), sc.data_ptr()
        arr[k].M, arr[k].C_out, arr[k].C_in, arr[k].C_in_real, arr[k].slices, arr[k].tiles_per_workgroup = M, co, ci, cr, sl, tw[k]
    return arr, keep, slices
def bench(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)
arr32, keep32, sl32 = build(six, False)
arr9, keep9, sl9 = build(six, True)
n = len(six)
out = {"fp32_us": bench(lambda: _lib.check(lib.bg_mlp_weight_grad_group(arr32, n, st), "fp32")), "split9_us": bench(lambda: _lib.check(lib.bg_mlp_weight_grad_group_split(arr9, n, 9, st), "split")),
       "slices_fp32": sl32, "slices_split": sl9}
print(json.dumps(out), flush=True)
for (co, ci, cr), (G, A, d32, _), (_, _, d9, _) in zip(six, keep32, keep9):
    ref = G.double().t() @ A.double()[:, :cr]
    row = {"layer": f"{co}x{ci}"}
    for name, d in (("fp32", d32), ("split9", d9)):
        e = d.double() - ref
        row[name] = {"rms": float(e.pow(2).mean().sqrt()), "mean_signed": float(e.mean()), "mean_signed_times_sign_of_result": float((e * ref.sign()).mean()),
                     "max": float(e.abs().max())}
    row["rms_ratio"] = round(row["split9"]["rms"] / row["fp32"]["rms"], 3)
    row["rms_of_result"] = float(ref.pow(2).mean().sqrt())
    print(json.dumps(row), flush=True)
