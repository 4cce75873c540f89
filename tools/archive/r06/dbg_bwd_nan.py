import sys, ctypes, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from booster_gym_amd import _lib
import test_gpu_mlp_chain_split_bwd as T
lib, st = _lib.load(), _lib.current_stream_ptr()
for M, dims, wgs in ((98304, (256, 128, 128), 0), (98304, (256, 256, 128), 0), (98304, (256, 128, 128), 96)):
    d, t = T._case(M, dims, 1, wgs)
    fin = _lib.ReduceProblem()
    for rep in range(2):
        t["G1"].fill_(float("nan")); t["G2"].fill_(float("nan"))
        _lib.check(lib.bg_mlp_chain_backward_split(ctypes.addressof(d), 1, fin, st))
        torch.cuda.synchronize()
        for name in ("G2", "G1"):
            y = t[name][:M]
            bad = ~torch.isfinite(y)
            if bad.any():
                idx = bad.nonzero()
                rows, cols = idx[:, 0], idx[:, 1]
                print(M, dims, wgs, "rep", rep, name, "bad", int(bad.sum()), "rows", rows.unique().numel(), "slabs", (rows // 128).unique()[:8].tolist(), "n slabs", (rows // 128).unique().numel(),
                      "row in slab", (rows % 128).unique()[:16].tolist(), "cols", cols.unique()[:40].tolist())
            else:
                r2 = (t["G3"].double() @ t["W3"].double()) * T._elup(t["A2"].double())
                ref = r2 if name == "G2" else (r2 @ t["W2"].double()) * T._elup(t["A1"].double())
                err = (y.double() - ref).abs()
                print(M, dims, wgs, "rep", rep, name, "finite; max err", err.max().item(), "cols of worst", (err.max(0).values > 1e-6).nonzero().flatten()[:20].tolist())
