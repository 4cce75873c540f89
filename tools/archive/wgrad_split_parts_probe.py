"""Split weight-gradient launch with its LDS-DMA copies removed from the loop (probe build, -DBG_PROBE_NO_STAGE): what is left is MFMA + split + LDS reads."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd import _lib
from booster_gym_amd.utils.model import plan_wgrad_slices
lib = _lib.load(); dev = "cuda:0"; st = _lib.current_stream_ptr()
ns = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libwgrad_split_nostage.bin"))
ns.bg_mlp_weight_grad_group_split.restype = C.c_int32
ns.bg_mlp_weight_grad_group_split.argtypes = [C.POINTER(_lib.WgradProblem), C.c_int32, C.c_int32, C.c_void_p]
M = 98304
def build(shapes):
    slices, tw = plan_wgrad_slices([(co, ci) for co, ci, _ in shapes], M, 256, share_rows=True)
    arr = (_lib.WgradProblem * len(shapes))(); keep = []
    for k, ((co, ci, cr), sl) in enumerate(zip(shapes, slices)):
        G = torch.randn(M, co, device=dev); A = torch.randn(M, ci, device=dev)
        dW = torch.empty(co, cr, device=dev); sc = torch.empty(sl * co * ci, device=dev); keep.append((G, A, dW, sc))
        arr[k].G, arr[k].A, arr[k].dW, arr[k].scratch = G.data_ptr(), A.data_ptr(), dW.data_ptr(), sc.data_ptr()
        arr[k].M, arr[k].C_out, arr[k].C_in, arr[k].C_in_real, arr[k].slices, arr[k].tiles_per_workgroup = M, co, ci, cr, sl, tw[k]
    return arr, keep, slices
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)
six = [(256, 64, 61), (256, 256, 256), (128, 256, 256), (256, 64, 47), (128, 256, 256), (128, 128, 128)]
for name, shapes in (("six layers", six), ("256x256 only", [(256, 256, 256)]), ("128x128 only", [(128, 128, 128)]), ("128x256 only", [(128, 256, 256)]), ("256x64 only", [(256, 64, 61)])):
    arr, keep, slices = build(shapes)
    n = len(shapes)
    out = {"case": name, "slices": slices, "gflop": round(sum(2.0 * M * co * ci for co, ci, _ in shapes) / 1e9, 1)}
    out["fp32_us"] = bench(lambda: lib.bg_mlp_weight_grad_group(arr, n, st))
    for t in (9, 6):
        out[f"split{t}_us"] = bench(lambda: lib.bg_mlp_weight_grad_group_split(arr, n, t, st))
        out[f"split{t}_no_copies_us"] = bench(lambda: ns.bg_mlp_weight_grad_group_split(arr, n, t, st))
    out["mfma9_us_at_2.2GHz"] = round(out["gflop"] * 1e9 * 9 / 2.5e15 * 2.4 / 2.2 * 1e6, 1)
    print(json.dumps(out), flush=True)
