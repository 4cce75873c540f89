"""Where does a wave of the fp32 weight-gradient launch spend its life?  Probe build of bg_wgrad.hip (-DBG_PROBE_TIMELINE) stamping the shader clock at
start of the row loop | end of the steady-state loop | end of the tail = start of the in-workgroup reduction | end (partial tile stored).  Six layers, M = 98,304."""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd import _lib
from booster_gym_amd.utils.model import plan_wgrad_slices
dev = "cuda:0"; st = _lib.current_stream_ptr(); _lib.load()
tl = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libwgrad_timeline.bin"))
tl.bg_mlp_weight_grad_group.restype = C.c_int32
tl.bg_mlp_weight_grad_group.argtypes = [C.POINTER(_lib.WgradProblem), C.c_int32, C.c_void_p]
tl.bg_probe_read_wgrad_timeline.argtypes = [C.c_void_p, C.c_size_t]
M = 98304
shapes = [(256, 64, 61), (256, 256, 256), (128, 256, 256), (256, 64, 47), (128, 256, 256), (128, 128, 128)]
slices, tw = plan_wgrad_slices([(co, ci) for co, ci, _ in shapes], M, 256, share_rows=False)
arr = (_lib.WgradProblem * len(shapes))(); keep = []
for k, ((co, ci, cr), sl) in enumerate(zip(shapes, slices)):
    G = torch.randn(M, co, device=dev); A = torch.randn(M, ci, device=dev)
    dW = torch.empty(co, cr, device=dev); sc = torch.empty(sl * co * ci, device=dev); keep.append((G, A, dW, sc))
    arr[k].G, arr[k].A, arr[k].dW, arr[k].scratch = G.data_ptr(), A.data_ptr(), dW.data_ptr(), sc.data_ptr()
    arr[k].M, arr[k].C_out, arr[k].C_in, arr[k].C_in_real, arr[k].slices, arr[k].tiles_per_workgroup = M, co, ci, cr, sl, tw[k]
for _ in range(3):
    assert tl.bg_mlp_weight_grad_group(arr, 6, st) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); tl.bg_mlp_weight_grad_group(arr, 6, st); e1.record(); torch.cuda.synchronize()
buf = np.zeros(1024 * 4 * 8, dtype=np.int64)
tl.bg_probe_read_wgrad_timeline(buf.ctypes.data, buf.nbytes)
nwg = sum(((co // 128) * max(1, ci // 128)) * s for (co, ci, _), s in zip(shapes, slices))
t = buf.reshape(1024, 4, 8)[:nwg].reshape(-1, 8).astype(np.float64)
d = np.diff(t[:, :4], axis=1)
rows_per_wave = M / (np.array(slices) * 4)
out = {"launch_pair_us": round(e0.elapsed_time(e1) * 1e3, 1), "workgroups": int(nwg), "slices": slices,
       "median_cycles": {"row loop": float(np.median(d[:, 0])), "tail": float(np.median(d[:, 1])), "reduction + store": float(np.median(d[:, 2]))},
       "p90_cycles": {"row loop": float(np.percentile(d[:, 0], 90)), "tail": float(np.percentile(d[:, 1], 90)), "reduction + store": float(np.percentile(d[:, 2], 90))},
       "max_cycles_row_loop": float(d[:, 0].max()), "max_cycles_total": float((t[:, 3] - t[:, 0]).max()),
       "mfma_cycles_per_wave_256x256": float(rows_per_wave[1] / 2 * 16 * 64), "rows_per_wave": [float(x) for x in rows_per_wave]}
print(json.dumps(out))
