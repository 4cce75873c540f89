"""What stands between a small kernel and the start of the next launch on the same stream?  (The update's mini-epoch shows 12-16 us in front of the
two kernels whose workgroups fill a CU -- the chained forward and the grouped weight gradients -- and 0-7 us in front of the others.)

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/probe/launch_gap.py ; python3 tools/probe/launch_gap.py --read OUT/.../kernel_trace.csv

Sequences, 40 repetitions each, host far ahead (everything is enqueued before the first kernel ends):
  A  fill -> chain (actor, one slab per workgroup)      B  fill -> chain (96 persistent workgroups)    C  fill -> per-layer kernel (36 KB of LDS)
  D  chain -> chain                                     E  fill -> fill                                F  per-layer -> chain
"""
import csv, ctypes as C, os, re, sys
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    def short(n):
        n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
        m = re.match(r"([\w:]+(<[^>]*>)?)", n)
        return (m.group(1) if m else n)[:40]
    gaps = {}
    for a, b in zip(rows, rows[1:]):
        key = (short(a["Kernel_Name"]), short(b["Kernel_Name"]), b["Grid_Size"] if "Grid_Size" in b else "")
        gaps.setdefault(key, []).append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
    for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
        if len(v) >= 10:
            v.sort()
            print(f"{k[0]:>40} -> {k[1]:<40} grid {k[2]:>8}: n {len(v):3d}  gap median {v[len(v)//2]:6.1f} us  (10 % {v[len(v)//10]:6.1f}, 90 % {v[9*len(v)//10]:6.1f})")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from booster_gym_amd import _lib
lib = _lib.load(); dev = "cuda:0"
torch.manual_seed(0)
M, dims = 98304, (64, 256, 128, 128)
K0, N1, N2, N3 = dims
x = torch.randn(M, K0, device=dev)
Ws = [torch.randn(n, k, device=dev) / k ** 0.5 for k, n in ((K0, N1), (N1, N2), (N2, N3))]
bs = [torch.randn(n, device=dev) * 0.1 for n in (N1, N2, N3)]
ys = [torch.empty(M, n, device=dev) for n in (N1, N2, N3)]
p = _lib.ptr
def chain_desc(wgs):
    return _lib.MlpChain(M, K0, N1, N2, N3, wgs, p(x), p(Ws[0]), p(bs[0]), p(Ws[1]), p(bs[1]), p(Ws[2]), p(bs[2]), p(ys[0]), p(ys[1]), p(ys[2]))
st = _lib.current_stream_ptr()
d0, d96 = chain_desc(0), chain_desc(96)
a0, a96 = (_lib.MlpChain * 1)(d0), (_lib.MlpChain * 1)(d96)
chain = lambda: _lib.check(lib.bg_mlp_chain_forward_group(C.addressof(a0), 1, st))
chain96 = lambda: _lib.check(lib.bg_mlp_chain_forward_group(C.addressof(a96), 1, st))
layer = lambda: _lib.check(lib.bg_mlp_layer_forward(M, N1, N2, p(ys[0]), p(Ws[1]), p(bs[1]), p(ys[1]), 1, st))
small = torch.zeros(1 << 16, device=dev)
small2 = torch.zeros(3 << 16, device=dev)  # (another grid size: the two fills are told apart by it)
fill = lambda: small.fill_(1.0)
fill2 = lambda: small2.fill_(2.0)
for f in (chain, chain96, layer, fill):
    f()
torch.cuda.synchronize()
R = 40
for name, seq in (("A", (fill, chain)), ("B", (fill, chain96)), ("C", (fill, layer)), ("D", (chain, chain)), ("E", (fill, fill2)), ("F", (layer, chain))):
    for _ in range(R):
        for f in seq:
            f()
    torch.cuda.synchronize()
print("done", flush=True)
