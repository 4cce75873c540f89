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
