"""Chained backward-data pass of a network's hidden layers on the bf16 matrix pipe with fp32 semantics (bg_mlp_chain_split_bwd.hip; the dX part of
`loss.backward()`, reference utils/runner.py:163 through utils/model.py:9-26):  G2 = (G3 W3) * elu'(A2), G1 = (G2 W2) * elu'(A1) and the bias gradients
(column sums of G2, G1).  Checked against float64 and against the fp32-MFMA layer kernels of the same op (bg_mlp_layer_backward, twice): the chain must be
at least as close to float64 (tolerance at the assertion: rms error <= 1.05 x, largest single error <= 2 x + 1e-7; column sums: largest error <= 1.5 x the
fp32 kernels' + 2e-5 of the largest sum)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _planes_t(w):
    """planes of W^T: n_out = W.shape[1], k_out = W.shape[0]"""
    from booster_gym_amd import _lib

    k, n = w.shape
    p = torch.zeros(2 * n * k * 3, dtype=torch.int16, device=DEV)   # (the planes of W^T, then of -W^T)
    _lib.check(_lib.load().bg_mlp_split_weights_pm(n, k, _lib.ptr(w), n, k, n, 1, _lib.ptr(p), _lib.current_stream_ptr()), "bg_mlp_split_weights_pm")
    return p


def _case(M, dims, seed, wgs=0, alternate=1):
    from booster_gym_amd import _lib

    N1, N2, N3 = dims
    g = torch.Generator(device="cpu").manual_seed(seed)
    G3 = (torch.randn(M, N3, generator=g) * 0.01).to(DEV)
    W3 = (torch.randn(N3, N2, generator=g) / N3**0.5).to(DEV)
    W2 = (torch.randn(N2, N1, generator=g) / N2**0.5).to(DEV)
    for W in (W3, W2):  # asymmetric entries catch transposed / permuted fragment maps
        W[3, 5] = 3.0; W[W.shape[0] - 1, 0] = -2.0
    pad = (M + 127) // 128 * 128
    # (the activations hold whole slabs, as the chained forward kernel leaves them: rows >= M finite)
    A2 = torch.zeros(pad, N2); A2[:M] = torch.nn.functional.elu(torch.randn(M, N2, generator=g)); A2 = A2.to(DEV)
    A1 = torch.zeros(pad, N1); A1[:M] = torch.nn.functional.elu(torch.randn(M, N1, generator=g)); A1 = A1.to(DEV)
    G2 = torch.full((pad, N2), float("nan"), device=DEV)
    G1 = torch.full((pad, N1), float("nan"), device=DEV)
    slabs = pad // 128
    part = torch.full((slabs * 4, N2 + N1), float("nan"), device=DEV)   # one record per (slab, wave)
    b2, b1 = torch.full((N2,), float("nan"), device=DEV), torch.full((N1,), float("nan"), device=DEV)
    P3, P2 = _planes_t(W3), _planes_t(W2)
    p = _lib.ptr
    d = _lib.MlpChainSplitBwd(M, N1, N2, N3, wgs, alternate, p(G3), p(P3), p(P2), p(A2), p(A1), p(G2), p(G1), p(part), p(b2), p(b1))
    return d, dict(G3=G3, W3=W3, W2=W2, A2=A2, A1=A1, G2=G2, G1=G1, part=part, b2=b2, b1=b1, P3=P3, P2=P2)


def _elup(a):
    return torch.where(a > 0, torch.ones_like(a), a + 1.0)


def _fp32_layers(M, t):
    """the same two layers through the fp32-MFMA layer kernel"""
    from booster_gym_amd import _lib

    lib, st, p = _lib.load(), _lib.current_stream_ptr(), _lib.ptr
    outs = []
    g = t["G3"]
    for W, A in ((t["W3"], t["A2"]), (t["W2"], t["A1"])):
        K, N = W.shape
        wt = W.t().contiguous()
        go, bg, sc = torch.empty(M, N, device=DEV), torch.empty(N, device=DEV), torch.empty(((M + 127) // 128) * N, device=DEV)
        _lib.check(lib.bg_mlp_layer_backward(M, K, N, p(g), p(wt), p(A), p(go), p(bg), p(sc), st), "bg_mlp_layer_backward")
        outs.append((go, bg))
        g = go
    return outs


def _run_and_check(M, dims, wgs, seed, alternate=1):
    from booster_gym_amd import _lib
    from booster_gym_amd.utils.utils import reduce_group

    lib, st = _lib.load(), _lib.current_stream_ptr()
    d, t = _case(M, dims, seed, wgs, alternate)
    fin = _lib.ReduceProblem()
    _lib.check(lib.bg_mlp_chain_backward_split(ctypes.addressof(d), 1, fin, st), "bg_mlp_chain_backward_split")
    reduce_group([fin])
    r2 = (t["G3"].double() @ t["W3"].double()) * _elup(t["A2"][:M].double())
    r1 = (r2 @ t["W2"].double()) * _elup(t["A1"][:M].double())
    f32 = _fp32_layers(M, t)
    stats = []
    for name, y, ref, (z, zb), b in (("G2", t["G2"], r2, f32[0], t["b2"]), ("G1", t["G1"], r1, f32[1], t["b1"])):
        assert torch.isfinite(y).all(), name
        assert (y[M:] == 0).all(), name  # rows of the last slab beyond M: zeros
        y = y[:M]
        err, err32 = (y.double() - ref).abs().max().item(), (z.double() - ref).abs().max().item()
        rms, rms32 = (y.double() - ref).pow(2).mean().sqrt().item(), (z.double() - ref).pow(2).mean().sqrt().item()
        # at least as close to float64 as the fp32-MFMA kernels: rms within 5 %, the largest single error (a tail statistic of a few ulps) within 2 x
        assert err <= 2.0 * err32 + 1e-7 and rms <= 1.05 * rms32 + 1e-9, (name, err, err32, rms, rms32)
        cs = ref.sum(0)
        cerr, cerr32 = (b.double() - cs).abs().max().item(), (zb.double() - cs).abs().max().item()
        # the column sums (bias gradients): with the alternating accumulation at the fp32 kernels' level (3 x + 1e-6 of the largest sum: both are a few
        # units of fp32 rounding of sums over up to 98,304 rows); accumulated the plain way the MFMA's rounding bias adds up over the rows (looser bound)
        slack = (3.0 * cerr32 + 1e-6 * cs.abs().max().item()) if alternate else (1.5 * cerr32 + 2e-5 * cs.abs().max().item())
        assert torch.isfinite(b).all() and cerr <= slack, (name, cerr, cerr32)
        stats.append((name, rms, rms32, err, err32, cerr, cerr32))
    return t, stats


@pytest.mark.parametrize("dims", [(256, 256, 128), (256, 128, 128)])
def test_split_chain_backward_gives_the_same_bits_however_the_slabs_are_dealt_to_workgroups(dims):
    """One slab per workgroup, the planner's shares, a count that gives some workgroups one slab and others two, an odd count: the same bits in G2, G1
    and (one record per (slab, wave), summed in a fixed order) in the bias gradients."""
    from booster_gym_amd import _lib
    from booster_gym_amd.utils.utils import reduce_group

    lib, st = _lib.load(), _lib.current_stream_ptr()
    M, ref = 384 * 128 - 57, None
    for wgs in (0, 168, 256, 37, 383):
        d, t = _case(M, dims, 31, wgs)
        fin = _lib.ReduceProblem()
        _lib.check(lib.bg_mlp_chain_backward_split(ctypes.addressof(d), 1, fin, st), "bg_mlp_chain_backward_split")
        reduce_group([fin])
        got = [t[k].clone() for k in ("G2", "G1", "b2", "b1")]
        assert all(torch.isfinite(x).all() for x in got)
        if ref is None:
            ref = got
        for k, (a, b) in enumerate(zip(ref, got)):
            assert torch.equal(a, b), (wgs, k, (a - b).abs().max().item(), int((a != b).sum()))


@pytest.mark.parametrize("M,dims,wgs", [(98304, (256, 128, 128), 0), (98304, (256, 256, 128), 0), (1000, (256, 256, 128), 0), (77, (256, 128, 128), 0),
                                        # persistent workgroups walking the slabs (the update's split of the CUs), a count that does not divide the slabs,
                                        # more workgroups than slabs
                                        (98304, (256, 128, 128), 96), (98304, (256, 256, 128), 160), (1000, (256, 256, 128), 3), (77, (256, 128, 128), 5)])
def test_split_chain_backward_matches_float64_as_well_as_the_fp32_layers(M, dims, wgs):
    t, stats = _run_and_check(M, dims, wgs, seed=M + dims[1] + wgs)
    print(f"split backward chain M={M} dims={dims} wgs={wgs}: (name, rms, rms fp32-MFMA, max, max fp32-MFMA, colsum err, colsum err fp32-MFMA) = {stats}")


def test_plain_accumulation_still_works_and_shows_the_bias_the_alternation_removes():
    t, plain = _run_and_check(98304, (256, 256, 128), 160, seed=9, alternate=0)
    t, alt = _run_and_check(98304, (256, 256, 128), 160, seed=9, alternate=1)
    print(f"column-sum error (G2, G1): plain {plain[0][5]:.2e} {plain[1][5]:.2e}, alternating {alt[0][5]:.2e} {alt[1][5]:.2e}, fp32 MFMA {alt[0][6]:.2e} {alt[1][6]:.2e}")
    assert alt[0][5] < 0.5 * plain[0][5] and alt[1][5] < 0.5 * plain[1][5]


def test_split_chain_backward_is_deterministic_groups_and_refusals():
    """Two runs give the same bits (fixed-order column sums); two networks in one launch; refusals."""
    from booster_gym_amd import _lib
    from booster_gym_amd.utils.utils import reduce_group

    lib, st = _lib.load(), _lib.current_stream_ptr()
    dc, tc = _case(98304, (256, 256, 128), 5, 160)
    da, ta = _case(2400, (256, 128, 128), 6, 0)
    arr = (_lib.MlpChainSplitBwd * 2)(dc, da)
    fins = (_lib.ReduceProblem * 2)()
    outs = []
    for _ in range(2):
        for t in (tc, ta):
            for k in ("G2", "G1", "b2", "b1", "part"):
                t[k].fill_(float("nan"))
        _lib.check(lib.bg_mlp_chain_backward_split(ctypes.addressof(arr), 2, fins, st), "bg_mlp_chain_backward_split")
        reduce_group([fins[0], fins[1]])
        outs.append([t[k].clone() for t in (tc, ta) for k in ("G2", "G1", "b2", "b1")])
    for x, y in zip(*outs):
        assert torch.isfinite(x[: 2400 if x.dim() == 2 and x.shape[0] < 98304 else None]).all() and torch.equal(x, y)
    ref = (ta["G3"].double() @ ta["W3"].double()) * _elup(ta["A2"][:2400].double())
    assert (ta["G2"][:2400].double() - ref).abs().max().item() < 1e-5
    fin = _lib.ReduceProblem()
    for field, val, rc in (("N2", 64, -4), ("M", 0, -1), ("G3", dc.G3 + 4, -1), ("PT2", None, -1), ("workgroups", -1, -1), ("bias_grad1", None, -1)):
        bad = _lib.MlpChainSplitBwd.from_buffer_copy(dc)
        setattr(bad, field, val)
        assert lib.bg_mlp_chain_backward_split(ctypes.addressof(bad), 1, fin, st) == rc, field
    assert lib.bg_mlp_chain_backward_split(ctypes.addressof(arr), 5, fins, st) == -1
