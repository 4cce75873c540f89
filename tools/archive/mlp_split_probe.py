"""Split-bf16 layer kernels (bg_mlp_layer_forward_split / _backward_split) against the fp32-MFMA kernels: error against a float64 reference
and time alone on the GPU, at the update's shapes (M = 98,304).  One JSON object per line."""
import json
import sys

import torch

sys.path.insert(0, ".")
from booster_gym_amd import _lib  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    M = 98304
    st = _lib.current_stream_ptr()
    for (K, N) in ((256, 256), (256, 128), (128, 128), (64, 256)):
        X = torch.randn(M, K, device=dev) * 0.7
        W = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev) * 0.1
        ref = torch.nn.functional.elu(X[:8192].double() @ W.double().t() + b.double())
        Y = torch.empty(M, N, device=dev)
        planes = torch.empty(N * K * 3, dtype=torch.int16, device=dev)
        _lib.check(lib.bg_mlp_split_weights(N, K, _lib.ptr(W), K, N, K, 0, _lib.ptr(planes), st), "split")
        out = {"shape": f"K={K} N={N}", "M": M}
        _lib.check(lib.bg_mlp_layer_forward(M, K, N, _lib.ptr(X), _lib.ptr(W), _lib.ptr(b), _lib.ptr(Y), 1, st), "fwd")
        out["fp32_mfma_max_err"] = float((Y[:8192].double() - ref).abs().max())
        out["fp32_mfma_rms_err"] = float((Y[:8192].double() - ref).pow(2).mean().sqrt())
        out["fp32_mfma_us"] = round(timeit(lambda: lib.bg_mlp_layer_forward(M, K, N, _lib.ptr(X), _lib.ptr(W), _lib.ptr(b), _lib.ptr(Y), 1, st)), 1)
        for terms in (9, 6):
            Y.zero_()
            _lib.check(lib.bg_mlp_layer_forward_split(M, K, N, _lib.ptr(X), _lib.ptr(planes), _lib.ptr(b), _lib.ptr(Y), 1, terms, st), "fwd split")
            out[f"split{terms}_max_err"] = float((Y[:8192].double() - ref).abs().max())
            out[f"split{terms}_rms_err"] = float((Y[:8192].double() - ref).pow(2).mean().sqrt())
            out[f"split{terms}_us"] = round(timeit(lambda: lib.bg_mlp_layer_forward_split(M, K, N, _lib.ptr(X), _lib.ptr(planes), _lib.ptr(b), _lib.ptr(Y), 1, terms, st)), 1)
        out["mfma_time_us_fp32_at_2.4GHz"] = round(2.0 * M * K * N / 157.3e12 * 1e6, 1)
        print(json.dumps(out), flush=True)
    # backward: Gout = (G Wt^T) * elu'(act), bias gradient = column sums
    for (K, N) in ((256, 256), (128, 256), (128, 128)):
        G = torch.randn(M, K, device=dev) * 1e-3
        Wup = torch.randn(K, N, device=dev) / K ** 0.5  # the upper layer's weight [K out][N in]
        Wt = Wup.t().contiguous()                       # [N][K]
        act = torch.nn.functional.elu(torch.randn(M, N, device=dev))
        d = torch.where(act[:8192] > 0, torch.ones_like(act[:8192]), act[:8192] + 1).double()
        ref = (G[:8192].double() @ Wup.double()) * d
        Gout = torch.empty(M, N, device=dev)
        bg = torch.empty(N, device=dev)
        scratch = torch.empty((M + 127) // 128 * N, device=dev)
        planes = torch.empty(N * K * 3, dtype=torch.int16, device=dev)
        _lib.check(lib.bg_mlp_split_weights(N, K, _lib.ptr(Wup), N, K, N, 1, _lib.ptr(planes), st), "split t")
        out = {"backward_shape": f"K={K} N={N}", "M": M}
        _lib.check(lib.bg_mlp_layer_backward(M, K, N, _lib.ptr(G), _lib.ptr(Wt), _lib.ptr(act), _lib.ptr(Gout), _lib.ptr(bg), _lib.ptr(scratch), st), "bwd")
        out["fp32_mfma_max_err"] = float((Gout[:8192].double() - ref).abs().max())
        bgref = Gout.double().sum(0)
        out["fp32_mfma_us"] = round(timeit(lambda: lib.bg_mlp_layer_backward(M, K, N, _lib.ptr(G), _lib.ptr(Wt), _lib.ptr(act), _lib.ptr(Gout), _lib.ptr(bg), _lib.ptr(scratch), st)), 1)
        for terms in (9, 6):
            Gout.zero_()
            _lib.check(lib.bg_mlp_layer_backward_split(M, K, N, _lib.ptr(G), _lib.ptr(planes), _lib.ptr(act), _lib.ptr(Gout), _lib.ptr(bg), _lib.ptr(scratch), terms, st), "bwd split")
            out[f"split{terms}_max_err"] = float((Gout[:8192].double() - ref).abs().max())
            out[f"split{terms}_bias_grad_rel_err"] = float(((bg.double() - Gout.double().sum(0)).abs().max() / bgref.abs().max()))
            out[f"split{terms}_us"] = round(timeit(lambda: lib.bg_mlp_layer_backward_split(M, K, N, _lib.ptr(G), _lib.ptr(planes), _lib.ptr(act), _lib.ptr(Gout), _lib.ptr(bg), _lib.ptr(scratch), terms, st)), 1)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
