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


# the six hidden-layer shapes of the two networks at the training batch (actor 47(64)-256-128-128, critic 61(64)-256-256-128; utils/model.py:9-26)
# + ragged / small / asymmetric cases: run lengths that are not multiples of the kernel's 16-row-pair trip, one slice group, wide C_in
@pytest.mark.parametrize("M,C_out,C_in,C_real,slices", [(98304, 256, 64, 47, 128), (98304, 128, 256, 256, 128), (98304, 128, 128, 128, 256),
                                                       (98304, 256, 64, 61, 128), (98304, 256, 256, 256, 64), (98304, 128, 256, 256, 64),
                                                       (1536, 128, 128, 128, 8), (1000, 256, 128, 128, 16), (70, 128, 64, 64, 8), (4098, 384, 256, 256, 24)])
def test_mlp_weight_grad_matches_torch_fp64(M, C_out, C_in, C_real, slices):
    """dW = G^T A against torch float64; the tolerance is that of torch's own fp32 GEMM of the same product (summation order only)."""
    from booster_gym_amd import _lib

    torch.manual_seed(M + C_out + 7 * C_in)
    G = torch.randn(M, C_out, device=DEV)
    A = torch.randn(M, C_in, device=DEV)
    A[:, C_real:] = 0.0  # zero-padded input columns, as the runner provides them
    # asymmetric markers: a transposed / permuted output map cannot pass
    G[:, 3] *= 3.0; A[:, 1] += 0.5; G[0, C_out - 1] = 40.0; A[0, C_real - 1] = -25.0
    dW = torch.full((C_out, C_real), float("nan"), device=DEV)
    scratch = torch.empty(slices * C_out * C_in, device=DEV)
    _lib.check(_lib.load().bg_mlp_weight_grad(M, C_out, C_in, C_real, _lib.ptr(G), _lib.ptr(A), _lib.ptr(dW), _lib.ptr(scratch), slices,
                                              _lib.current_stream_ptr()), "bg_mlp_weight_grad")
    ref64 = G.double().t() @ A.double()[:, :C_real]
    ref32 = G.t() @ A[:, :C_real]
    err, err_t = (dW.double() - ref64).abs().max().item(), (ref32.double() - ref64).abs().max().item()
    assert torch.isfinite(dW).all() and err <= max(4 * err_t, 1e-4), (err, err_t)
    # deterministic: a second launch gives the same bits
    dW2 = torch.empty_like(dW)
    _lib.check(_lib.load().bg_mlp_weight_grad(M, C_out, C_in, C_real, _lib.ptr(G), _lib.ptr(A), _lib.ptr(dW2), _lib.ptr(scratch), slices,
                                              _lib.current_stream_ptr()), "bg_mlp_weight_grad")
    assert torch.equal(dW, dW2)


def test_mlp_weight_grad_rejects_bad_arguments():
    from booster_gym_amd import _lib

    lib = _lib.load()
    G = torch.zeros(256, 128, device=DEV); A = torch.zeros(256, 128, device=DEV); dW = torch.zeros(128, 128, device=DEV); sc = torch.zeros(8 * 128 * 128, device=DEV)
    st = _lib.current_stream_ptr()
    assert lib.bg_mlp_weight_grad(256, 100, 128, 128, _lib.ptr(G), _lib.ptr(A), _lib.ptr(dW), _lib.ptr(sc), 8, st) == -4   # C_out
    assert lib.bg_mlp_weight_grad(256, 128, 96, 96, _lib.ptr(G), _lib.ptr(A), _lib.ptr(dW), _lib.ptr(sc), 8, st) == -4     # C_in
    assert lib.bg_mlp_weight_grad(255, 128, 128, 128, _lib.ptr(G), _lib.ptr(A), _lib.ptr(dW), _lib.ptr(sc), 8, st) == -4   # odd M
    assert lib.bg_mlp_weight_grad(256, 128, 128, 128, _lib.ptr(G), _lib.ptr(A), _lib.ptr(dW), _lib.ptr(sc), 12, st) == -4  # slices % 8
    assert lib.bg_mlp_weight_grad(32, 128, 128, 128, _lib.ptr(G), _lib.ptr(A), _lib.ptr(dW), _lib.ptr(sc), 8, st) == -4    # slices * 8 > M
    assert lib.bg_mlp_weight_grad(256, 128, 128, 129, _lib.ptr(G), _lib.ptr(A), _lib.ptr(dW), _lib.ptr(sc), 8, st) == -1   # C_in_real
    assert lib.bg_mlp_weight_grad(256, 128, 128, 128, _lib.ptr(G), None, _lib.ptr(dW), _lib.ptr(sc), 8, st) == -1


@pytest.mark.parametrize("share_rows", [True, False])
def test_mlp_weight_grad_group_matches_torch_fp64(share_rows):
    """bg_mlp_weight_grad_group: the six hidden-layer weight gradients of both networks in one launch pair, slices sized by plan_wgrad_slices
    (uneven on purpose: 12 / 23 / 24 slices, runs that are not multiples of the kernel's 16-row-pair trip); each against torch float64."""
    from booster_gym_amd import _lib
    from booster_gym_amd.utils.model import plan_wgrad_slices

    M = 98304
    shapes = [(256, 64, 61), (256, 256, 256), (128, 256, 256), (256, 64, 47), (128, 256, 256), (128, 128, 128)]
    slices, tw = plan_wgrad_slices([(co, ci) for co, ci, _ in shapes], M, 256, share_rows=share_rows)
    assert sum((co // 128) * max(1, ci // 128) // w * s for (co, ci, _), s, w in zip(shapes, slices, tw)) <= 256 and min(slices) >= 8
    assert tw == ([2, 4, 2, 2, 2, 1] if share_rows else [1] * 6)
    torch.manual_seed(5)
    arr = (_lib.WgradProblem * len(shapes))()
    keep = []
    for k, ((co, ci, cr), sl) in enumerate(zip(shapes, slices)):
        G = torch.randn(M, co, device=DEV); A = torch.randn(M, ci, device=DEV); A[:, cr:] = 0.0
        G[:, 3] *= 3.0; A[:, 1] += 0.5; G[0, co - 1] = 40.0; A[0, cr - 1] = -25.0
        dW = torch.full((co, cr), float("nan"), device=DEV); sc = torch.empty(sl * co * ci, device=DEV)
        keep.append((G, A, dW, sc))
        arr[k].G, arr[k].A, arr[k].dW, arr[k].scratch = G.data_ptr(), A.data_ptr(), dW.data_ptr(), sc.data_ptr()
        arr[k].M, arr[k].C_out, arr[k].C_in, arr[k].C_in_real, arr[k].slices, arr[k].tiles_per_workgroup = M, co, ci, cr, sl, tw[k]
    _lib.check(_lib.load().bg_mlp_weight_grad_group(arr, len(shapes), _lib.current_stream_ptr()), "bg_mlp_weight_grad_group")
    for (co, ci, cr), (G, A, dW, sc) in zip(shapes, keep):
        ref64 = G.double().t() @ A.double()[:, :cr]
        ref32 = G.t() @ A[:, :cr]
        err, err_t = (dW.double() - ref64).abs().max().item(), (ref32.double() - ref64).abs().max().item()
        assert torch.isfinite(dW).all() and err <= max(4 * err_t, 1e-4), (co, ci, err, err_t)
    first = [dW.clone() for _, _, dW, _ in keep]
    _lib.check(_lib.load().bg_mlp_weight_grad_group(arr, len(shapes), _lib.current_stream_ptr()), "bg_mlp_weight_grad_group")
    assert all(torch.equal(a, dW) for a, (_, _, dW, _) in zip(first, keep))  # deterministic
    assert _lib.load().bg_mlp_weight_grad_group(arr, 9, _lib.current_stream_ptr()) == -1
    arr[0].slices = M  # more than M / 8
    assert _lib.load().bg_mlp_weight_grad_group(arr, len(shapes), _lib.current_stream_ptr()) == -4
    arr[0].slices, arr[5].tiles_per_workgroup = slices[0], 2  # the 128 x 128 layer has one tile
    assert _lib.load().bg_mlp_weight_grad_group(arr, len(shapes), _lib.current_stream_ptr()) == -4


@pytest.mark.parametrize("M", [96, 768, 2400])
def test_mlp_weight_grad_group_small_batches(M):
    """The grouped weight-gradient launch at small batches (4, 32 and 100 envs x 24 steps), where (M / 2) / 16 runs of 16 row pairs do not
    cover slices x waves and some waves get an EMPTY run: such a wave must add zeros (and, when its run starts at row 0, must not read in
    front of the arrays: its clamped prologue load now lands on row 0).  Operands sit at the very start of their own allocations, next to
    canaries, so that a read before G or A would be a read of another buffer or of unmapped memory.  Against torch float64, both layouts."""
    from booster_gym_amd import _lib
    from booster_gym_amd.utils.model import plan_wgrad_slices

    shapes = [(256, 64, 61), (256, 256, 256), (128, 256, 256), (256, 64, 47), (128, 256, 256), (128, 128, 128)]
    torch.manual_seed(7)
    for share_rows in (False, True):
        slices, tw = plan_wgrad_slices([(co, ci) for co, ci, _ in shapes], M, 256, share_rows=share_rows)
        assert all(sl >= 1 and sl * (4 // w) <= max(M // 32, 4 // w) for sl, w in zip(slices, tw)), (slices, tw)
        arr = (_lib.WgradProblem * len(shapes))()
        keep = []
        for k, ((co, ci, cr), sl) in enumerate(zip(shapes, slices)):
            G = torch.randn(M, co, device=DEV); A = torch.randn(M, ci, device=DEV); A[:, cr:] = 0.0
            dW = torch.full((co, cr), float("nan"), device=DEV); sc = torch.empty(sl * co * ci, device=DEV)
            keep.append((G, A, dW, sc))
            arr[k].G, arr[k].A, arr[k].dW, arr[k].scratch = G.data_ptr(), A.data_ptr(), dW.data_ptr(), sc.data_ptr()
            arr[k].M, arr[k].C_out, arr[k].C_in, arr[k].C_in_real, arr[k].slices, arr[k].tiles_per_workgroup = M, co, ci, cr, sl, tw[k]
        _lib.check(_lib.load().bg_mlp_weight_grad_group(arr, len(shapes), _lib.current_stream_ptr()), "bg_mlp_weight_grad_group")
        torch.cuda.synchronize()
        for (co, ci, cr), (G, A, dW, sc) in zip(shapes, keep):
            ref64 = G.double().t() @ A.double()[:, :cr]
            err, err_t = (dW.double() - ref64).abs().max().item(), ((G.t() @ A[:, :cr]).double() - ref64).abs().max().item()
            assert torch.isfinite(dW).all() and err <= max(4 * err_t, 1e-4), (M, share_rows, co, ci, err, err_t)


def _chain_case(M, dims, seed):
    import ctypes
    from booster_gym_amd import _lib

    K0, N1, N2, N3 = dims
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(M, K0, generator=g).to(DEV)
    Ws = [(torch.randn(n, k, generator=g) / k**0.5).to(DEV) for k, n in ((K0, N1), (N1, N2), (N2, N3))]
    for W in Ws:  # asymmetric entries catch transposed / permuted fragment maps
        W[3, 5] = 3.0; W[W.shape[0] - 1, 0] = -2.0
    bs = [(torch.randn(n, generator=g) * 0.3).to(DEV) for n in (N1, N2, N3)]
    pad = (M + 127) // 128 * 128
    ys = [torch.full((pad, n), float("nan"), device=DEV) for n in (N1, N2, N3)]
    p = _lib.ptr
    d = _lib.MlpChain(M, K0, N1, N2, N3, 0, p(x), p(Ws[0]), p(bs[0]), p(Ws[1]), p(bs[1]), p(Ws[2]), p(bs[2]), p(ys[0]), p(ys[1]), p(ys[2]), None, None, None)
    return d, x, Ws, bs, ys


def test_mlp_chain_forward_value_head():
    """The optional scalar output layer of bg_mlp_chain_forward_group (the critic's values, taken from the registers that hold the last activations)
    against fp64 on every row, ragged batch included; the activations themselves are unchanged by it; half a descriptor is refused."""
    import ctypes
    from booster_gym_amd import _lib

    lib, st, p = _lib.load(), _lib.current_stream_ptr(), _lib.ptr
    for M in (102400, 1000):
        d, x, Ws, bs, ys = _chain_case(M, (64, 256, 256, 128), seed=11 + M)
        g = torch.Generator(device="cpu").manual_seed(5)
        vw, vb = (torch.randn(128, generator=g) * 0.1).to(DEV), torch.randn(1, generator=g).to(DEV)
        vo = torch.full((M,), float("nan"), device=DEV)
        d.v_w, d.v_b, d.v_out = p(vw), p(vb), p(vo)
        _lib.check(lib.bg_mlp_chain_forward_group(ctypes.addressof(d), 1, st))
        _chain_check(M, x, Ws, bs, ys)
        ref = ys[2][:M].double() @ vw.double() + vb.double()
        assert torch.isfinite(vo).all() and (vo.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    d.v_b = None
    assert lib.bg_mlp_chain_forward_group(ctypes.addressof(d), 1, st) == -1 and b"value head" in lib.bg_last_error()


def _chain_check(M, x, Ws, bs, ys):
    """against torch fp64 on every row, and bit for bit against three per-layer launches (the same sums in the same order)"""
    from booster_gym_amd import _lib

    lib, st = _lib.load(), _lib.current_stream_ptr()
    ref, hin = x.double(), x
    for l in range(3):
        ref = torch.nn.functional.elu(ref @ Ws[l].double().t() + bs[l].double())
        y = ys[l][:M]
        assert torch.isfinite(y).all(), l
        err = (y.double() - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (l, err)
        z = torch.empty(M, Ws[l].shape[0], device=DEV)
        _lib.check(lib.bg_mlp_layer_forward(M, hin.shape[1], Ws[l].shape[0], _lib.ptr(hin), _lib.ptr(Ws[l]), _lib.ptr(bs[l]), _lib.ptr(z), 1, st))
        assert torch.equal(y, z), l
        hin = z


@pytest.mark.parametrize("M,dims,wgs", [(98304, (64, 256, 128, 128), 0), (102400, (64, 256, 256, 128), 0), (1000, (64, 256, 256, 128), 0), (77, (64, 256, 128, 128), 0),
                                        # bg_mlp_chain::workgroups > 0: that many workgroups walk the slabs (the update's split of the CUs between the two
                                        # networks' launches: 160 x 5 critic slabs, 96 x 8 actor slabs); a count that does not divide the slabs; more
                                        # workgroups than slabs (= one per slab)
                                        (98304, (64, 256, 128, 128), 96), (102400, (64, 256, 256, 128), 160), (1000, (64, 256, 256, 128), 3), (77, (64, 256, 128, 128), 5)])
def test_mlp_chain_forward_matches_torch_fp64(M, dims, wgs):
    """bg_mlp_chain_forward_group (three Linear+ELU layers of a network in one launch, activations in registers between them) at the training
    shapes (M = 98,304 / 102,400) and on ragged batches: every stored activation against torch fp64 and the per-layer kernels."""
    import ctypes
    from booster_gym_amd import _lib

    d, x, Ws, bs, ys = _chain_case(M, dims, seed=M + dims[2])
    d.workgroups = wgs
    _lib.check(_lib.load().bg_mlp_chain_forward_group(ctypes.addressof(d), 1, _lib.current_stream_ptr()))
    _chain_check(M, x, Ws, bs, ys)


def test_mlp_chain_forward_group_of_two_networks_and_bad_arguments():
    import ctypes
    from booster_gym_amd import _lib

    lib, st = _lib.load(), _lib.current_stream_ptr()
    dc, xc, Wc, bc, yc = _chain_case(2400 + 4096, (64, 256, 256, 128), seed=1)
    da, xa, Wa, ba, ya = _chain_case(2400, (64, 256, 128, 128), seed=2)
    arr = (_lib.MlpChain * 2)(dc, da)
    _lib.check(lib.bg_mlp_chain_forward_group(ctypes.addressof(arr), 2, st))
    _chain_check(2400 + 4096, xc, Wc, bc, yc)
    _chain_check(2400, xa, Wa, ba, ya)
    # the flat entry point is the group of one
    for y in ya:
        y.fill_(float("nan"))
    p = _lib.ptr
    _lib.check(lib.bg_mlp_chain_forward(2400, 64, 256, 128, 128, p(xa), p(Wa[0]), p(ba[0]), p(Wa[1]), p(ba[1]), p(Wa[2]), p(ba[2]), p(ya[0]), p(ya[1]), p(ya[2]), st))
    _chain_check(2400, xa, Wa, ba, ya)
    assert lib.bg_mlp_chain_forward(2400, 64, 512, 128, 128, p(xa), p(Wa[0]), p(ba[0]), p(Wa[1]), p(ba[1]), p(Wa[2]), p(ba[2]), p(ya[0]), p(ya[1]), p(ya[2]), st) == -4
    assert lib.bg_mlp_chain_forward(2400, 47, 256, 128, 128, p(xa), p(Wa[0]), p(ba[0]), p(Wa[1]), p(ba[1]), p(Wa[2]), p(ba[2]), p(ya[0]), p(ya[1]), p(ya[2]), st) == -4
    assert lib.bg_mlp_chain_forward(0, 64, 256, 128, 128, p(xa), p(Wa[0]), p(ba[0]), p(Wa[1]), p(ba[1]), p(Wa[2]), p(ba[2]), p(ya[0]), p(ya[1]), p(ya[2]), st) == -1
    assert lib.bg_mlp_chain_forward(2400, 64, 256, 128, 128, p(xa).value + 4, p(Wa[0]), p(ba[0]), p(Wa[1]), p(ba[1]), p(Wa[2]), p(ba[2]), p(ya[0]), p(ya[1]), p(ya[2]), st) == -1
    assert lib.bg_mlp_chain_forward_group(ctypes.addressof(arr), 5, st) == -1
    assert b"widths" in lib.bg_last_error() or b"network" in lib.bg_last_error()
