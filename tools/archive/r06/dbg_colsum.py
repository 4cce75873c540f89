import sys, ctypes, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from booster_gym_amd import _lib
from booster_gym_amd.utils.utils import reduce_group
import test_gpu_mlp_chain_split_bwd as T
lib, st = _lib.load(), _lib.current_stream_ptr()
M, dims = 98304, (256, 128, 128)
d, t = T._case(M, dims, 1, 0)
fin = _lib.ReduceProblem()
_lib.check(lib.bg_mlp_chain_backward_split(ctypes.addressof(d), 1, fin, st))
reduce_group([fin])
torch.cuda.synchronize()
G2 = t["G2"][:M].double()
own = G2.sum(0)
print("b2 vs fp64 sum of the chain's own G2: max err", (t["b2"].double() - own).abs().max().item(), "max |sum|", own.abs().max().item())
part = t["part"].double()[:, :dims[1]]
slab = G2.view(768, 128, dims[1]).sum(1)
e = (part - slab).abs()
print("per-record err max", e.max().item(), "mean", e.mean().item(), "typical |record|", slab.abs().mean().item())
k = e.argmax().item(); print("worst record/feature", k // dims[1], k % dims[1], part.flatten()[k].item(), slab.flatten()[k].item())
# which rows are missing / duplicated?  least squares: difference as a combination of rows
rec, f = k // dims[1], k % dims[1]
diff = (part[rec] - slab[rec])
rows = G2[rec * 128 : rec * 128 + 128]
sol = torch.linalg.lstsq(rows.t(), diff.unsqueeze(1)).solution.flatten()
print("lstsq row weights (nonzero => row counted wrongly):", [(i, round(v, 3)) for i, v in enumerate(sol.tolist()) if abs(v) > 0.05][:20])
r2 = (t["G3"].double() @ t["W3"].double()) * T._elup(t["A2"].double())
f32 = T._fp32_layers(M, t)
for name, y in (("split chain", t["G2"][:M]), ("fp32 MFMA layer", f32[0][0])):
    err = y.double() - r2
    ulp = torch.abs(r2) * 2.0 ** -23
    print(name, "mean signed err", err.mean().item(), "mean err*sign(ref)", (err * torch.sign(r2)).mean().item(), "rms", err.pow(2).mean().sqrt().item(),
          "mean err in ulps of |ref|", (err / ulp.clamp_min(1e-30)).mean().item(), "toward-zero part in ulps", (err * torch.sign(r2) / ulp.clamp_min(1e-30)).mean().item())
