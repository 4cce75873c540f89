"""Time the chained split-bf16 backward kernel (bg_mlp_chain_backward_split) against the two fp32-MFMA layer launches it replaces, at the update's
shapes: each network alone and the pair side by side on two streams.   python tools/chain_split_bwd_probe.py [reps]"""
import ctypes, json, sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from booster_gym_amd import _lib
import test_gpu_mlp_chain_split_bwd as T

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
lib, p = _lib.load(), _lib.ptr
nets = {"critic": (98304, (256, 256, 128)), "actor": (98304, (256, 128, 128))}
cases = {}
for name, (M, dims) in nets.items():
    d, t = T._case(M, dims, 3, 0)
    wts = [t["W3"].t().contiguous(), t["W2"].t().contiguous()]
    tmp = dict(g2=torch.empty(M, dims[1], device=T.DEV), g1=torch.empty(M, dims[0], device=T.DEV), sc=torch.empty(768 * 256, device=T.DEV), b=torch.empty(256, device=T.DEV))
    cases[name] = dict(d=d, t=t, wts=wts, tmp=tmp, M=M, dims=dims, flop=2.0 * M * (dims[2] * dims[1] + dims[1] * dims[0]))


def run_split(c):
    fin = _lib.ReduceProblem()
    _lib.check(lib.bg_mlp_chain_backward_split(ctypes.addressof(c["d"]), 1, fin, _lib.current_stream_ptr()))


def run_fp32(c):
    st, t, tmp, (N1, N2, N3), M = _lib.current_stream_ptr(), c["t"], c["tmp"], c["dims"], c["M"]
    _lib.check(lib.bg_mlp_layer_backward(M, N3, N2, p(t["G3"]), p(c["wts"][0]), p(t["A2"]), p(tmp["g2"]), p(tmp["b"]), p(tmp["sc"]), st))
    _lib.check(lib.bg_mlp_layer_backward(M, N2, N1, p(tmp["g2"]), p(c["wts"][1]), p(t["A1"]), p(tmp["g1"]), p(tmp["b"]), p(tmp["sc"]), st))


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


out = {}
for name, c in cases.items():
    us_f = timed(lambda: run_fp32(c), reps)
    for wg in (0, 96, 160, 168, 256):
        c["d"].workgroups = wg
        us_s = timed(lambda: run_split(c), reps)
        out[f"{name}_alone_wg{wg}"] = {"split9_us": round(us_s, 1), "fp32_layers_us": round(us_f, 1), "split9_TFs_fp32_equiv": round(c["flop"] / us_s / 1e6, 1)}
        print(name, wg, out[f"{name}_alone_wg{wg}"], flush=True)
side = torch.cuda.Stream()


def pair(split):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        (run_split if split else run_fp32)(cases["critic"])
    (run_split if split else run_fp32)(cases["actor"])
    main.wait_stream(side)


for wc, wa in ((168, 88), (160, 96), (176, 80), (0, 0)):
    cases["critic"]["d"].workgroups, cases["actor"]["d"].workgroups = wc, wa
    us_s, us_f = timed(lambda: pair(True), reps), timed(lambda: pair(False), reps)
    fl = cases["critic"]["flop"] + cases["actor"]["flop"]
    out[f"pair_{wc}_{wa}"] = {"split9_us": round(us_s, 1), "fp32_layers_us": round(us_f, 1), "split9_TFs_fp32_equiv": round(fl / us_s / 1e6, 1)}
    print("pair", wc, wa, out[f"pair_{wc}_{wa}"], flush=True)
print(json.dumps(out))
