"""Where does a wave of the fp32-MFMA layer kernel spend its life?  Probe build of bg_mlp.hip (-DBG_PROBE_TIMELINE) stamping the shader clock at the phase
boundaries of every wave: prologue (first loads, first LDS stage, barrier) | own work on each PAIR of 32-deep chunks (stamp before the barrier) | epilogue
(bias + ELU + quad transpose + stores, by column tile).  Forward layers at M = 98,304; medians over all waves, in shader cycles."""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd import _lib
dev = "cuda:0"; st = _lib.current_stream_ptr(); _lib.load()
tl = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libmlp_timeline.bin"))
tl.bg_mlp_layer_forward.restype = C.c_int32
tl.bg_mlp_layer_forward.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
tl.bg_probe_read_timeline.argtypes = [C.c_void_p, C.c_size_t]
M = 98304
for K, N in ((256, 256), (256, 128), (128, 128), (64, 256)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.06; b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
    for _ in range(3):
        tl.bg_mlp_layer_forward(M, K, N, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), 1, st)
    torch.cuda.synchronize()
    nwg = (M // 128) * (N // 128)
    buf = np.zeros(2048 * 4 * 16, dtype=np.int64)
    tl.bg_probe_read_timeline(buf.ctypes.data, buf.nbytes)
    t = buf.reshape(2048, 4, 16)[:nwg].reshape(-1, 16).astype(np.float64)
    P = K // 64  # chunk pairs
    names = ["prologue"] + [f"chunk pair {c}" for c in range(P)] + ["last barrier", "epilogue"]
    d = np.diff(t[:, : P + 4], axis=1)
    out = {"shape": f"K={K} N={N}", "waves": int(t.shape[0]), "wave_lifetime_median": float(np.median(t[:, P + 3] - t[:, 0])),
           "mfma_cycles_per_chunk_pair_per_wave": 2 * 64 * 64,
           "median_cycles": {n: float(np.median(d[:, k])) for k, n in enumerate(names)}}
    e0 = t[:, P + 2]
    out["epilogue_by_tile_median"] = [float(np.median(t[:, 12] - e0)), float(np.median(t[:, 13] - t[:, 12])), float(np.median(t[:, 14] - t[:, 13])),
                                      float(np.median(t[:, P + 3] - t[:, 14]))]
    print(json.dumps(out), flush=True)
