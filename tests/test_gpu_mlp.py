"""Hand-written fp32-MFMA MLP layer kernels vs the plain PyTorch fp32 reference of the same op (torch.addmm + ELU).
fp32 MFMA is an exact-fp32 fma chain; differences come from the summation order only (tolerance 2e-5 relative to the row norm)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("M,K,N,elu", [(98304, 256, 256, True), (102400, 256, 128, True), (98304, 128, 128, True), (4096, 256, 256, False), (1000, 128, 128, True),
                                       (130, 256, 128, True), (4096, 64, 384, True)])
def test_mlp_layer_forward_matches_torch(M, K, N, elu):
    from booster_gym_amd import _lib

    torch.manual_seed(M + K + N)
    x = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) * (1.0 / K**0.5)
    # asymmetric weights + non-trivial bias catch transposed / permuted fragment maps
    w[3, 5] = 7.0; w[N - 1, 0] = -3.0
    b = torch.randn(N, device=DEV)
    y = torch.full((M, N), float("nan"), device=DEV)
    _lib.check(_lib.load().bg_mlp_layer_forward(M, K, N, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), int(elu), _lib.current_stream_ptr()))
    ref = torch.addmm(b, x, w.t())
    if elu:
        ref = torch.nn.functional.elu(ref)
    ref64 = torch.addmm(b.double(), x.double(), w.double().t())
    if elu:
        ref64 = torch.nn.functional.elu(ref64)
    assert torch.isfinite(y).all()
    err = (y.double() - ref64).abs().max().item()
    err_torch = (ref.double() - ref64).abs().max().item()
    assert err <= max(4 * err_torch, 1e-5), (err, err_torch)


def test_mlp_layer_forward_rejects_unsupported_shapes():
    from booster_gym_amd import _lib

    x = torch.zeros(128, 61, device=DEV); w = torch.zeros(256, 61, device=DEV); b = torch.zeros(256, device=DEV); y = torch.zeros(128, 256, device=DEV)
    rc = _lib.load().bg_mlp_layer_forward(128, 61, 256, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), 1, _lib.current_stream_ptr())
    assert rc == -4 and b"unsupported" in _lib.load().bg_last_error()


@pytest.mark.parametrize("M,K,N", [(98304, 256, 256), (98304, 128, 256), (98304, 128, 128), (1000, 256, 128), (130, 128, 384)])
def test_mlp_layer_backward_matches_torch(M, K, N):
    """Gout = (G W) * elu'(act_below), bias_grad_below = column sums; W is [K][N] in torch layout (out = K, in = N), the kernel takes W^T."""
    from booster_gym_amd import _lib

    torch.manual_seed(M + 3 * K + N)
    G = torch.randn(M, K, device=DEV)
    W = torch.randn(K, N, device=DEV) * (1.0 / K**0.5)
    W[2, 7] = 5.0; W[K - 1, 0] = -4.0
    z = torch.randn(M, N, device=DEV)
    act = torch.nn.functional.elu(z)
    ref = (G.double() @ W.double()) * torch.where(z > 0, torch.ones_like(z), act + 1.0).double()
    Wt = W.t().contiguous()
    out = torch.full((M, N), float("nan"), device=DEV)
    bg = torch.zeros(N, device=DEV)
    scratch = torch.empty(((M + 127) // 128) * N, device=DEV)
    _lib.check(_lib.load().bg_mlp_layer_backward(M, K, N, _lib.ptr(G), _lib.ptr(Wt), _lib.ptr(act), _lib.ptr(out), _lib.ptr(bg), _lib.ptr(scratch),
                                                 _lib.current_stream_ptr()))
    ref32 = (G @ W) * torch.where(z > 0, torch.ones_like(z), act + 1.0)
    err, err_t = (out.double() - ref).abs().max().item(), (ref32.double() - ref).abs().max().item()
    assert torch.isfinite(out).all() and err <= max(4 * err_t, 1e-5), (err, err_t)
    cs = ref.sum(0)
    assert torch.allclose(bg.double(), cs, rtol=1e-4, atol=2e-3 * max(1.0, cs.abs().max().item()))
