"""Chained forward of a network's three Linear + ELU layers on the bf16 matrix pipe with fp32 semantics (bg_mlp_chain_split.hip; reference
utils/model.py:9-26 under utils/runner.py:132,147): every fp32 operand as the exact sum of three bf16 numbers, all 9 cross products accumulated in fp32.
Checked against float64 and against the fp32-MFMA chain of the same op (bg_mlp_chain_forward_group): the split chain must be at least as close to
float64 as the fp32-MFMA chain (tolerance written at the assertion: rms error <= 1.05 x, largest single error <= 2 x + 1e-7)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _planes(w, n_out, k_out):
    """the planes of W and, behind them, of -W (bg_mlp_split_weights_pm)"""
    from booster_gym_amd import _lib

    p = torch.zeros(2 * n_out * k_out * 3, dtype=torch.int16, device=DEV)
    _lib.check(_lib.load().bg_mlp_split_weights_pm(n_out, k_out, _lib.ptr(w), w.shape[1], w.shape[0], w.shape[1], 0, _lib.ptr(p), _lib.current_stream_ptr()),
               "bg_mlp_split_weights_pm")
    return p


def _case(M, dims, seed, k_real=None, wgs=0, alternate=1):
    from booster_gym_amd import _lib

    K0, N1, N2, N3 = dims
    k_real = k_real or K0
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.zeros(M, K0)
    x[:, :k_real] = torch.randn(M, k_real, generator=g)
    x = x.to(DEV)
    Ws = [(torch.randn(n, k, generator=g) / k**0.5).to(DEV) for k, n in ((k_real, N1), (N1, N2), (N2, N3))]
    for W in Ws:  # asymmetric entries catch transposed / permuted fragment maps
        W[3, 5] = 3.0; W[W.shape[0] - 1, 0] = -2.0
    bs = [(torch.randn(n, generator=g) * 0.3).to(DEV) for n in (N1, N2, N3)]
    pad = (M + 127) // 128 * 128
    ys = [torch.full((pad, n), float("nan"), device=DEV) for n in (N1, N2, N3)]
    Ps = [_planes(Ws[0], N1, K0), _planes(Ws[1], N2, N1), _planes(Ws[2], N3, N2)]
    p = _lib.ptr
    d = _lib.MlpChainSplit(M, K0, N1, N2, N3, wgs, alternate, 0, p(x), p(Ps[0]), p(Ps[1]), p(Ps[2]), p(bs[0]), p(bs[1]), p(bs[2]), p(ys[0]), p(ys[1]), p(ys[2]), None, None, None)
    return d, x, Ws, bs, ys, Ps


def _fp32_chain(M, dims, x, Ws, bs):
    """the fp32-MFMA chain on the same inputs (first layer's weights zero-padded to K0 columns)"""
    from booster_gym_amd import _lib

    K0, N1, N2, N3 = dims
    w0 = torch.zeros(N1, K0, device=DEV); w0[:, : Ws[0].shape[1]] = Ws[0]
    pad = (M + 127) // 128 * 128
    zs = [torch.empty(pad, n, device=DEV) for n in (N1, N2, N3)]
    p = _lib.ptr
    d = _lib.MlpChain(M, K0, N1, N2, N3, 0, p(x), p(w0), p(bs[0]), p(Ws[1]), p(bs[1]), p(Ws[2]), p(bs[2]), p(zs[0]), p(zs[1]), p(zs[2]), None, None, None)
    _lib.check(_lib.load().bg_mlp_chain_forward_group(ctypes.addressof(d), 1, _lib.current_stream_ptr()), "bg_mlp_chain_forward_group")
    return zs


def _check(M, dims, x, Ws, bs, ys):
    zs = _fp32_chain(M, dims, x, Ws, bs)
    ref = x.double()[:, : Ws[0].shape[1]]
    out = []
    for l in range(3):
        ref = torch.nn.functional.elu(ref @ Ws[l].double().t() + bs[l].double())
        y, z = ys[l][:M], zs[l][:M]
        assert torch.isfinite(y).all(), l
        err, err32 = (y.double() - ref).abs().max().item(), (z.double() - ref).abs().max().item()
        rms, rms32 = (y.double() - ref).pow(2).mean().sqrt().item(), (z.double() - ref).pow(2).mean().sqrt().item()
        # at least as close to float64 as the fp32-MFMA chain: rms within 5 %, the largest single error (a tail statistic of a few ulps) within 2 x
        assert err <= 2.0 * err32 + 1e-7 and rms <= 1.05 * rms32 + 1e-9, (l, err, err32, rms, rms32)
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (l, err)
        out.append((rms, rms32, err, err32))
    return out


@pytest.mark.parametrize("M,dims,k_real,wgs", [(98304, (64, 256, 128, 128), 47, 0), (102400, (64, 256, 256, 128), 61, 0), (1000, (64, 256, 256, 128), 61, 0),
                                               (77, (64, 256, 128, 128), 47, 0),
                                               # persistent workgroups walking the slabs (the update's split of the CUs: 160 x 5 critic, 96 x 8 actor slabs),
                                               # a count that does not divide the slabs, more workgroups than slabs
                                               (98304, (64, 256, 128, 128), 47, 96), (102400, (64, 256, 256, 128), 61, 160), (1000, (64, 256, 256, 128), 64, 3),
                                               (77, (64, 256, 128, 128), 64, 5)])
def test_split_chain_forward_matches_float64_as_well_as_the_fp32_chain(M, dims, k_real, wgs):
    from booster_gym_amd import _lib

    d, x, Ws, bs, ys, Ps = _case(M, dims, seed=M + dims[2], k_real=k_real, wgs=wgs)
    _lib.check(_lib.load().bg_mlp_chain_forward_split(ctypes.addressof(d), 1, _lib.current_stream_ptr()), "bg_mlp_chain_forward_split")
    stats = _check(M, dims, x, Ws, bs, ys)
    print(f"split chain M={M} dims={dims}: (rms, rms fp32-MFMA, max, max fp32-MFMA) per layer = {stats}")


def test_alternating_the_sign_of_the_accumulation_removes_the_bias():
    """The bf16 MFMA's accumulator does not round to nearest: measured, a POSITIVE accumulated value comes out low and a negative one unbiased, so
    accumulated the plain way every output carries a bias of one sign (mean signed error 7 % of the rms error here) that everything summed over rows
    downstream collects.  With `alternate` odd slabs accumulate the negated sums: same exact products, same rms error, and the bias becomes
    sign-symmetric (toward zero, half the size): it cancels in sums of values of both signs -- the gradients; tests/test_gpu_mlp_chain_split_bwd.py
    shows the column sums at the fp32 kernels' level -- and is halved in the mean of these all-but-positive ELU outputs.  Also: the planes of -W are
    the planes of W with the sign bits flipped."""
    from booster_gym_amd import _lib

    M, dims = 98304, (64, 256, 256, 128)
    out = {}
    for alt in (0, 1):
        d, x, Ws, bs, ys, Ps = _case(M, dims, seed=5, k_real=61, alternate=alt)
        _lib.check(_lib.load().bg_mlp_chain_forward_split(ctypes.addressof(d), 1, _lib.current_stream_ptr()), "bg_mlp_chain_forward_split")
        ref = x.double()[:, :61]
        for l in range(3):
            ref = torch.nn.functional.elu(ref @ Ws[l].double().t() + bs[l].double())
        err = ys[2][:M].double() - ref
        out[alt] = (err.mean().abs().item(), err.pow(2).mean().sqrt().item())
        half = Ps[1].numel() // 2
        assert torch.equal(Ps[1][:half] ^ -32768, Ps[1][half:])   # (int16 view: xor with the sign bit)
    zs = _fp32_chain(M, dims, x, Ws, bs)
    e32 = zs[2][:M].double() - ref
    bias32, rms32 = e32.mean().abs().item(), e32.pow(2).mean().sqrt().item()
    (bias0, rms0), (bias1, rms1) = out[0], out[1]
    print(f"layer 3 outputs, |mean signed error| / rms: plain {bias0:.2e} / {rms0:.2e}, alternating {bias1:.2e} / {rms1:.2e}, fp32 MFMA {bias32:.2e} / {rms32:.2e}")
    assert rms1 <= 1.05 * rms0 and rms1 <= 1.05 * rms32
    assert bias1 <= 0.6 * bias0 and bias0 > 5.0 * bias32


@pytest.mark.parametrize("dims,k_real", [((64, 256, 256, 128), 61), ((64, 256, 128, 128), 47)])
def test_split_chain_forward_gives_the_same_bits_however_the_slabs_are_dealt_to_workgroups(dims, k_real):
    """Which workgroup walks which slab -- one slab each, the planner's shares, a count that gives some workgroups ONE slab and others two (first slab =
    last slab: prologue and tail with nothing between), an odd count -- changes no bit of the three activations or of the value head's output."""
    from booster_gym_amd import _lib

    lib, st, p = _lib.load(), _lib.current_stream_ptr(), _lib.ptr
    M = 400 * 128 - 57
    g = torch.Generator(device="cpu").manual_seed(9)
    vw, vb = (torch.randn(128, generator=g) * 0.1).to(DEV), torch.randn(1, generator=g).to(DEV)
    ref = None
    for wgs in (0, 160, 256, 33, 399):
        d, x, Ws, bs, ys, Ps = _case(M, dims, seed=21, k_real=k_real, wgs=wgs)
        vo = torch.full((M,), float("nan"), device=DEV)
        d.v_w, d.v_b, d.v_out = p(vw), p(vb), p(vo)
        _lib.check(lib.bg_mlp_chain_forward_split(ctypes.addressof(d), 1, st), "bg_mlp_chain_forward_split")
        got = [y[:M].clone() for y in ys] + [vo.clone()]
        assert all(torch.isfinite(t).all() for t in got)
        if ref is None:
            ref = got
        for k, (a, b) in enumerate(zip(ref, got)):
            assert torch.equal(a, b), (wgs, k, (a - b).abs().max().item(), int((a != b).sum()))


def test_split_chain_value_head_group_and_bad_arguments():
    """The scalar output layer taken from the registers (the critic's values) on every row of a ragged batch; two networks in one launch; the same
    slabs give the same bits whether they run in one launch or in pieces (the rollout evaluates each step's rows as soon as they exist); refusals."""
    from booster_gym_amd import _lib

    lib, st, p = _lib.load(), _lib.current_stream_ptr(), _lib.ptr
    M = 2400 + 4096
    dc, xc, Wc, bc, yc, Pc = _case(M, (64, 256, 256, 128), seed=1, k_real=61)
    da, xa, Wa, ba, ya, Pa = _case(2400, (64, 256, 128, 128), seed=2, k_real=47)
    g = torch.Generator(device="cpu").manual_seed(5)
    vw, vb = (torch.randn(128, generator=g) * 0.1).to(DEV), torch.randn(1, generator=g).to(DEV)
    vo = torch.full((M,), float("nan"), device=DEV)
    dc.v_w, dc.v_b, dc.v_out = p(vw), p(vb), p(vo)
    arr = (_lib.MlpChainSplit * 2)(dc, da)
    _lib.check(lib.bg_mlp_chain_forward_split(ctypes.addressof(arr), 2, st), "bg_mlp_chain_forward_split")
    _check(M, (64, 256, 256, 128), xc, Wc, bc, yc)
    _check(2400, (64, 256, 128, 128), xa, Wa, ba, ya)
    ref = yc[2][:M].double() @ vw.double() + vb.double()
    assert torch.isfinite(vo).all() and (vo.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    # rows [1280, 1280 + 2560) alone: the same bits in the same places
    keep = [y.clone() for y in yc] + [vo.clone()]
    for y in yc:
        y[1280 : 1280 + 2560].fill_(float("nan"))
    vo[1280 : 1280 + 2560].fill_(float("nan"))
    part = _lib.MlpChainSplit.from_buffer_copy(dc)
    part.M = 2560
    part.X = dc.X + 4 * 1280 * 64
    part.Y1, part.Y2, part.Y3 = dc.Y1 + 4 * 1280 * 256, dc.Y2 + 4 * 1280 * 256, dc.Y3 + 4 * 1280 * 128
    part.v_out = dc.v_out + 4 * 1280
    _lib.check(lib.bg_mlp_chain_forward_split(ctypes.addressof(part), 1, st), "bg_mlp_chain_forward_split")
    for a, b in zip(keep, yc + [vo]):
        assert torch.equal(a[:M], b[:M])
    # refusals
    dc.v_b = None
    assert lib.bg_mlp_chain_forward_split(ctypes.addressof(dc), 1, st) == -1 and b"value head" in lib.bg_last_error()
    dc.v_b = p(vb)
    for field, val, rc in (("N1", 512, -4), ("K0", 47, -4), ("M", 0, -1), ("X", dc.X + 4, -1), ("P2", None, -1), ("workgroups", -1, -1)):
        bad = _lib.MlpChainSplit.from_buffer_copy(dc)
        setattr(bad, field, val)
        assert lib.bg_mlp_chain_forward_split(ctypes.addressof(bad), 1, st) == rc, field
    assert lib.bg_mlp_chain_forward_split(ctypes.addressof(arr), 5, st) == -1
