import sys, ctypes, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from booster_gym_amd import _lib
import test_gpu_mlp_chain_split_bwd as T
lib, st = _lib.load(), _lib.current_stream_ptr()
M, dims, wgs = 98304, (256, 128, 128), 96
d, t = T._case(M, dims, 1, wgs)
fin = _lib.ReduceProblem()
_lib.check(lib.bg_mlp_chain_backward_split(ctypes.addressof(d), 1, fin, st))
torch.cuda.synchronize()
r2 = (t["G3"].double() @ t["W3"].double()) * T._elup(t["A2"].double())
r1 = (r2 @ t["W2"].double()) * T._elup(t["A1"].double())
for name, y, ref in (("G2", t["G2"][:M], r2), ("G1", t["G1"][:M], r1)):
    err = (y.double() - ref).abs()
    bad = err > 1e-5
    idx = bad.nonzero()
    rows, cols = idx[:, 0], idx[:, 1]
    slab = rows // 128
    print(name, "bad", int(bad.sum()), "slabs: n", slab.unique().numel(), "slab // 96 (index in workgroup)", (slab // 96).unique().tolist(), "slab % 96 first", (slab % 96).unique()[:10].tolist())
    print("   row in slab", (rows % 128).unique().tolist()[:40], "n", (rows % 128).unique().numel())
    print("   cols", cols.unique().tolist())
    k = idx[0]
    print("   example", k.tolist(), y[k[0], k[1]].item(), ref[k[0], k[1]].item(), "ratio", (y[k[0], k[1]].double() / ref[k[0], k[1]]).item())
    # is the wrong value the right product with another row's / tile's elu' factor?
    un = (t["G3"].double() @ t["W3"].double()) if name == "G2" else (r2 @ t["W2"].double())
    fac = y[k[0], k[1]].double() / un[k[0], k[1]]
    print("   implied factor", fac.item(), "true factor", T._elup(t["A2" if name == "G2" else "A1"].double())[k[0], k[1]].item())
