"""Time the chained split-bf16 forward kernel (bg_mlp_chain_forward_split) against the fp32-MFMA chain (bg_mlp_chain_forward_group) at the update's
shapes: each network alone, and the pair side by side on two streams with the CU split of the update (160 critic + 96 actor workgroups).
Usage: python tools/chain_split_probe.py [reps]"""
import ctypes
import json
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from booster_gym_amd import _lib  # noqa: E402
import test_gpu_mlp_chain_split as T  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
lib = _lib.load()
nets = {"critic": (102400, (64, 256, 256, 128), 61, 160), "actor": (98304, (64, 256, 128, 128), 47, 96)}
cases = {}
for name, (M, dims, kr, wg) in nets.items():
    d, x, Ws, bs, ys, Ps = T._case(M, dims, seed=3, k_real=kr)
    w0 = torch.zeros(dims[1], dims[0], device=T.DEV); w0[:, :kr] = Ws[0]
    p = _lib.ptr
    zs = [torch.empty_like(y) for y in ys]
    f = _lib.MlpChain(M, *dims, 0, p(x), p(w0), p(bs[0]), p(Ws[1]), p(bs[1]), p(Ws[2]), p(bs[2]), p(zs[0]), p(zs[1]), p(zs[2]), None, None, None)
    cases[name] = dict(d=d, f=f, keep=(x, Ws, bs, ys, Ps, w0, zs), wg=wg, flop=2.0 * M * (kr * dims[1] + dims[1] * dims[2] + dims[2] * dims[3]))


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


side = torch.cuda.Stream()
out = {}
for name, c in cases.items():
    for wg in (0, c["wg"], 256):
        c["d"].workgroups = c["f"].workgroups = wg
        st = _lib.current_stream_ptr()
        us_s = timed(lambda: _lib.check(lib.bg_mlp_chain_forward_split(ctypes.addressof(c["d"]), 1, st)), reps)
        us_f = timed(lambda: _lib.check(lib.bg_mlp_chain_forward_group(ctypes.addressof(c["f"]), 1, st)), reps)
        out[f"{name}_alone_wg{wg}"] = {"split9_us": round(us_s, 1), "fp32_us": round(us_f, 1), "split9_TFs_fp32_equiv": round(c["flop"] / us_s / 1e6, 1)}
        print(name, wg, out[f"{name}_alone_wg{wg}"], flush=True)


def pair(split):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    c, a = cases["critic"], cases["actor"]
    with torch.cuda.stream(side):
        st = _lib.current_stream_ptr()
        if split:
            _lib.check(lib.bg_mlp_chain_forward_split(ctypes.addressof(c["d"]), 1, st))
        else:
            _lib.check(lib.bg_mlp_chain_forward_group(ctypes.addressof(c["f"]), 1, st))
    st = _lib.current_stream_ptr()
    if split:
        _lib.check(lib.bg_mlp_chain_forward_split(ctypes.addressof(a["d"]), 1, st))
    else:
        _lib.check(lib.bg_mlp_chain_forward_group(ctypes.addressof(a["f"]), 1, st))
    main.wait_stream(side)


for wc, wa in ((160, 96), (168, 88), (152, 104), (144, 112)):
    cases["critic"]["d"].workgroups = cases["critic"]["f"].workgroups = wc
    cases["actor"]["d"].workgroups = cases["actor"]["f"].workgroups = wa
    us_s, us_f = timed(lambda: pair(True), reps), timed(lambda: pair(False), reps)
    fl = cases["critic"]["flop"] + cases["actor"]["flop"]
    out[f"pair_{wc}_{wa}"] = {"split9_us": round(us_s, 1), "fp32_us": round(us_f, 1), "split9_TFs_fp32_equiv": round(fl / us_s / 1e6, 1)}
    print("pair", wc, wa, out[f"pair_{wc}_{wa}"], flush=True)
print(json.dumps(out))
