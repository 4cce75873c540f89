"""Split-bf16 form of the fused layer kernels (bg_mlp_split.hip, opt-in BG_GEMM_SPLIT): every fp32 operand as the exact sum of three bf16
numbers, products on the bf16 MFMA pipe, fp32 accumulation.  Checked against float64 and against the fp32-MFMA kernel of the same op: with all 9
cross terms the result must be at least as close to float64 as the fp32-MFMA kernel's (tolerance: rms error <= 1.05 x its rms error, largest
single error <= 2 x its largest -- a tail statistic of a few ulps of the largest outputs); with 6 terms rms within 2 x.  The planes themselves are checked bit for bit: hi + mid + lo == w exactly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _planes(w, n_out, k_out, transpose=False):
    from booster_gym_amd import _lib

    p = torch.empty(n_out * k_out * 3, dtype=torch.int16, device=DEV)
    _lib.check(_lib.load().bg_mlp_split_weights(n_out, k_out, _lib.ptr(w), w.shape[1], w.shape[0], w.shape[1], int(transpose), _lib.ptr(p),
                                                _lib.current_stream_ptr()), "bg_mlp_split_weights")
    return p


def _unpack(p, n_out, k_out):
    """planes [n][k / 32][3][32] bf16 with the k permutation of split_planes_kernel -> three float64 arrays [n][k]"""
    raw = p.cpu().numpy().view(np.uint16).reshape(n_out, k_out // 32, 3, 32)
    f = (raw.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    k = np.arange(32)
    s, h, q = k >> 3, (k >> 2) & 1, k & 3
    pos = (s >> 1) * 16 + h * 8 + (s & 1) * 4 + q
    return [f[:, :, pl, :][:, :, pos].reshape(n_out, k_out) for pl in range(3)]


@pytest.mark.parametrize("n,k,src_cols,transpose", [(256, 64, 47, False), (256, 256, 256, False), (128, 256, 256, True), (128, 128, 128, True)])
def test_split_planes_are_an_exact_decomposition(n, k, src_cols, transpose):
    torch.manual_seed(n + k)
    w = (torch.randn(k if transpose else n, n if transpose else src_cols, device=DEV) * torch.logspace(-6, 3, n if transpose else src_cols, device=DEV))
    w[0, 0] = 0.0; w[1, 1] = 1.0; w[2, 2] = -3.4e38; w[3, 3] = 1.1754944e-38 * 300  # zero, exact bf16, near the top and near the bottom of the range
    hi, mid, lo = _unpack(_planes(w, n, k, transpose), n, k)
    want = (w.t() if transpose else w).double().cpu().numpy()
    full = np.zeros((n, k)); full[:, : want.shape[1]] = want
    assert np.array_equal(hi + mid + lo, full)  # exact, not approximate; padded columns are zero
    assert np.all(np.abs(mid) <= np.abs(hi) * 2.0 ** -7 + 1e-300) and np.all(np.abs(lo) <= np.abs(hi) * 2.0 ** -15 + 1e-300)


@pytest.mark.parametrize("terms", [9, 6])
@pytest.mark.parametrize("M,K,N,k_real,elu", [(98304, 256, 256, 256, True), (102400, 256, 128, 256, True), (98304, 128, 128, 128, True), (98304, 64, 256, 47, True),
                                              (1000, 128, 128, 128, True), (130, 256, 128, 256, False), (4096, 64, 384, 61, True)])
def test_split_forward_matches_float64_as_well_as_the_fp32_kernel(M, K, N, k_real, elu, terms):
    from booster_gym_amd import _lib

    lib, st = _lib.load(), _lib.current_stream_ptr()
    torch.manual_seed(M + K + N)
    x = torch.zeros(M, K, device=DEV); x[:, :k_real] = torch.randn(M, k_real, device=DEV)
    w = torch.randn(N, k_real, device=DEV) * (1.0 / k_real**0.5)
    w[3, 5] = 7.0; w[N - 1, 0] = -3.0
    b = torch.randn(N, device=DEV)
    y = torch.full((M, N), float("nan"), device=DEV)
    _lib.check(lib.bg_mlp_layer_forward_split(M, K, N, _lib.ptr(x), _lib.ptr(_planes(w, N, K)), _lib.ptr(b), _lib.ptr(y), int(elu), terms, st))
    wpad = torch.zeros(N, K, device=DEV); wpad[:, :k_real] = w
    y32 = torch.empty(M, N, device=DEV)
    _lib.check(lib.bg_mlp_layer_forward(M, K, N, _lib.ptr(x), _lib.ptr(wpad), _lib.ptr(b), _lib.ptr(y32), int(elu), st))
    ref = torch.addmm(b.double(), x.double(), wpad.double().t())
    if elu:
        ref = torch.nn.functional.elu(ref)
    assert torch.isfinite(y).all()
    err, err32 = (y.double() - ref).abs().max().item(), (y32.double() - ref).abs().max().item()
    rms, rms32 = (y.double() - ref).pow(2).mean().sqrt().item(), (y32.double() - ref).pow(2).mean().sqrt().item()
    slack = 1.05 if terms == 9 else 2.0
    assert err <= 2.0 * err32 + 1e-7 and rms <= slack * rms32 + 1e-9, (err, err32, rms, rms32)


@pytest.mark.parametrize("terms", [9, 6])
@pytest.mark.parametrize("M,K,N", [(98304, 256, 256), (98304, 128, 256), (98304, 128, 128), (1000, 256, 128), (130, 128, 384)])
def test_split_backward_matches_float64_as_well_as_the_fp32_kernel(M, K, N, terms):
    from booster_gym_amd import _lib

    lib, st = _lib.load(), _lib.current_stream_ptr()
    torch.manual_seed(M + 3 * K + N)
    G = torch.randn(M, K, device=DEV)
    W = torch.randn(K, N, device=DEV) * (1.0 / K**0.5)
    W[2, 7] = 5.0; W[K - 1, 0] = -4.0
    z = torch.randn(M, N, device=DEV)
    act = torch.nn.functional.elu(z)
    ref = (G.double() @ W.double()) * torch.where(z > 0, torch.ones_like(z), act + 1.0).double()
    out, out32 = torch.full((M, N), float("nan"), device=DEV), torch.empty(M, N, device=DEV)
    bg, bg32 = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    scratch = torch.empty(((M + 127) // 128) * N, device=DEV)
    _lib.check(lib.bg_mlp_layer_backward_split(M, K, N, _lib.ptr(G), _lib.ptr(_planes(W, N, K, transpose=True)), _lib.ptr(act), _lib.ptr(out), _lib.ptr(bg),
                                               _lib.ptr(scratch), terms, st))
    Wt = W.t().contiguous()
    _lib.check(lib.bg_mlp_layer_backward(M, K, N, _lib.ptr(G), _lib.ptr(Wt), _lib.ptr(act), _lib.ptr(out32), _lib.ptr(bg32), _lib.ptr(scratch), st))
    err, err32 = (out.double() - ref).abs().max().item(), (out32.double() - ref).abs().max().item()
    rms, rms32 = (out.double() - ref).pow(2).mean().sqrt().item(), (out32.double() - ref).pow(2).mean().sqrt().item()
    slack = 1.05 if terms == 9 else 2.0
    assert torch.isfinite(out).all() and err <= 2.0 * err32 + 1e-7 and rms <= slack * rms32 + 1e-9, (err, err32, rms, rms32)
    cs = ref.sum(0)
    assert torch.allclose(bg.double(), cs, rtol=1e-4, atol=2e-3 * max(1.0, cs.abs().max().item()))


def test_split_entries_reject_bad_arguments():
    from booster_gym_amd import _lib

    lib, st = _lib.load(), _lib.current_stream_ptr()
    x = torch.zeros(128, 128, device=DEV); w = torch.zeros(128, 128, device=DEV); b = torch.zeros(128, device=DEV); y = torch.zeros(128, 128, device=DEV)
    p = _planes(w, 128, 128)
    assert lib.bg_mlp_layer_forward_split(128, 128, 128, _lib.ptr(x), _lib.ptr(p), _lib.ptr(b), _lib.ptr(y), 1, 5, st) == -4 and b"terms" in lib.bg_last_error()
    assert lib.bg_mlp_layer_forward_split(128, 96, 128, _lib.ptr(x), _lib.ptr(p), _lib.ptr(b), _lib.ptr(y), 1, 9, st) == -4
    assert lib.bg_mlp_layer_forward_split(128, 128, 100, _lib.ptr(x), _lib.ptr(p), _lib.ptr(b), _lib.ptr(y), 1, 9, st) == -4
    assert lib.bg_mlp_split_weights(128, 100, _lib.ptr(w), 128, 128, 128, 0, _lib.ptr(p), st) == -1  # k_out must be a multiple of 32
    assert lib.bg_mlp_layer_backward_split(128, 64, 128, _lib.ptr(x), _lib.ptr(p), _lib.ptr(x), _lib.ptr(y), _lib.ptr(b), _lib.ptr(y), 9, st) == -4


@pytest.mark.parametrize("terms", [9, 6])
def test_full_update_in_split_mode_matches_reference_loop(terms):
    """Runner.update() with the layer kernels in split mode against the reference loop restated (oracle/ppo_ref.py), same bounds as the fp32 test."""
    from booster_gym_amd.utils.config import load_cfg
    from booster_gym_amd.utils.model import ActorCritic, MLPTrainer
    from booster_gym_amd.utils.runner import Runner
    from oracle.ppo_ref import ppo_update_reference

    old = MLPTrainer.SPLIT
    MLPTrainer.SPLIT = terms
    try:
        n, E = 128, 3
        cfg = load_cfg("T1", {"env.num_envs": n, "terrain.type": "plane", "runner.mini_epochs": E})
        r = Runner(cfg=cfg)
        obs, infos = r.env.reset()
        r.buffer["obses"][0].copy_(obs); r.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
        r.rollout()
        T = cfg["runner"]["horizon_length"]
        ref_model = ActorCritic(12, 47, 14).to(DEV)
        ref_model.load_state_dict(r.model.state_dict())
        b = r.buffer
        stats_ref, lr_ref = ppo_update_reference(ref_model, torch.optim.Adam(ref_model.parameters(), lr=1e-5), b["obses"][:T].clone(),
                                                 b["privileged_obses"][:T].clone(), b["actions"].clone(), b["rewards"].clone(), b["dones"].clone(),
                                                 b["time_outs"].clone(), b["obses"][T].clone(), b["privileged_obses"][T].clone(), mini_epochs=E,
                                                 learning_rate=1e-5)
        summ = r._summarize(r.update())
        assert r._actor_tr.planes[0] is not None and r._critic_tr.planes_t[1] is not None  # the split kernels did run
        for (k, p), (k2, q) in zip(r.model.named_parameters(), ref_model.named_parameters()):
            assert k == k2 and torch.allclose(p, q, rtol=1e-3, atol=2e-6), (k, (p - q).abs().max().item())
        for k in ("value_loss", "actor_loss", "bound_loss", "entropy", "kl_mean"):
            assert abs(summ[k] - stats_ref[k]) <= 2e-4 * max(1.0, abs(stats_ref[k])), (k, summ[k], stats_ref[k])
        assert abs(summ["lr"] - lr_ref) < 1e-9
    finally:
        MLPTrainer.SPLIT = old


@pytest.mark.parametrize("terms", [9, 6])
@pytest.mark.parametrize("M", [98304, 4096 + 32 * 7])
def test_split_weight_grad_group_matches_float64_as_well_as_the_fp32_kernel(M, terms):
    """bg_mlp_weight_grad_group_split: the six hidden-layer weight gradients in one launch pair, rows shared through LDS (tiles_per_workgroup = the
    layer's tile count), against torch float64 and beside the fp32-MFMA launch on the same inputs; deterministic; refusals."""
    from booster_gym_amd import _lib
    from booster_gym_amd.utils.model import plan_wgrad_slices

    lib, st = _lib.load(), _lib.current_stream_ptr()
    shapes = [(256, 64, 61), (256, 256, 256), (128, 256, 256), (256, 64, 47), (128, 256, 256), (128, 128, 128)]
    slices, tw = plan_wgrad_slices([(co, ci) for co, ci, _ in shapes], M, 256, share_rows=True)
    torch.manual_seed(5)
    arr = (_lib.WgradProblem * len(shapes))()
    keep = []
    for k, ((co, ci, cr), sl) in enumerate(zip(shapes, slices)):
        G = torch.randn(M, co, device=DEV); A = torch.randn(M, ci, device=DEV); A[:, cr:] = 0.0
        G[:, 3] *= 3.0; A[:, 1] += 0.5; G[0, co - 1] = 40.0; A[0, cr - 1] = -25.0; G[M - 1, 0] = 17.0; A[M - 1, 0] = 2.0
        dW = torch.full((co, cr), float("nan"), device=DEV); sc = torch.empty(sl * co * ci, device=DEV)
        keep.append((G, A, dW, sc))
        arr[k].G, arr[k].A, arr[k].dW, arr[k].scratch = G.data_ptr(), A.data_ptr(), dW.data_ptr(), sc.data_ptr()
        arr[k].M, arr[k].C_out, arr[k].C_in, arr[k].C_in_real, arr[k].slices, arr[k].tiles_per_workgroup = M, co, ci, cr, sl, tw[k]
    _lib.check(lib.bg_mlp_weight_grad_group(arr, len(shapes), st), "bg_mlp_weight_grad_group")
    fp32 = [dW.clone() for _, _, dW, _ in keep]
    for _, _, dW, _ in keep:
        dW.fill_(float("nan"))
    _lib.check(lib.bg_mlp_weight_grad_group_split(arr, len(shapes), terms, st), "bg_mlp_weight_grad_group_split")
    for (co, ci, cr), (G, A, dW, sc), d32 in zip(shapes, keep, fp32):
        ref64 = G.double().t() @ A.double()[:, :cr]
        rms, rms32 = (dW.double() - ref64).pow(2).mean().sqrt().item(), (d32.double() - ref64).pow(2).mean().sqrt().item()
        err, err32 = (dW.double() - ref64).abs().max().item(), (d32.double() - ref64).abs().max().item()
        # sums over 98,304 rows: the rounding of the ACCUMULATOR dominates here, not the products.  The bf16 MFMA's accumulator truncates, which left one
        # negative offset on every element (0.7 of the rms error, round 6: profiles/r06_wgrad_split_error_parts.jsonl) until the sub-ranges of the batch
        # took turns accumulating the negated sums: with 9 products the split launch is now at or below the fp32 kernel's rms error (it was 1.2 x), and
        # its mean signed error is a small fraction of its rms error; 6 products drop terms of 2^-24 of a product and stay in a band
        slack = 1.05 if terms == 9 else 2.5
        bias = (dW.double() - ref64).mean().abs().item()
        print(f"split weight gradients {co}x{ci}, {terms} products: rms {rms:.3e} (fp32 MFMA {rms32:.3e}), |mean signed error| {bias:.2e}, max {err:.2e} ({err32:.2e})")
        assert torch.isfinite(dW).all() and rms <= slack * rms32 + 1e-7 and err <= 2.5 * err32 + 1e-6, (co, ci, rms, rms32, err, err32)
        if terms == 9 and M >= 98304:
            assert bias <= 0.2 * rms, (co, ci, bias, rms)
    first = [dW.clone() for _, _, dW, _ in keep]
    _lib.check(lib.bg_mlp_weight_grad_group_split(arr, len(shapes), terms, st), "bg_mlp_weight_grad_group_split")
    assert all(torch.equal(a, dW) for a, (_, _, dW, _) in zip(first, keep))  # deterministic
    assert lib.bg_mlp_weight_grad_group_split(arr, len(shapes), 4, st) == -4
    arr[1].tiles_per_workgroup = 1  # the 256 x 256 layer's four tiles must share a workgroup here
    assert lib.bg_mlp_weight_grad_group_split(arr, len(shapes), terms, st) == -4 and b"tiles_per_workgroup" in lib.bg_last_error()
    arr[1].tiles_per_workgroup, arr[1].M = 4, M + 16
    assert lib.bg_mlp_weight_grad_group_split(arr, len(shapes), terms, st) == -4 and b"multiple of 32" in lib.bg_last_error()


def test_yaml_switch_selects_split_mode():
    """parallel.gemm_split: 9 / 6 turn the split GEMMs on for the process (like BG_GEMM_SPLIT), anything else but 0 is refused."""
    from booster_gym_amd.utils.config import load_cfg
    from booster_gym_amd.utils.model import MLPTrainer
    from booster_gym_amd.utils.runner import Runner

    old = MLPTrainer.SPLIT
    try:
        Runner(cfg=load_cfg("T1", {"env.num_envs": 64, "terrain.type": "plane", "parallel.gemm_split": 6}))
        assert MLPTrainer.SPLIT == 6
        with pytest.raises(ValueError, match="gemm_split"):
            Runner(cfg=load_cfg("T1", {"env.num_envs": 64, "terrain.type": "plane", "parallel.gemm_split": 3}))
    finally:
        MLPTrainer.SPLIT = old
