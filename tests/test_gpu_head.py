"""Fused output-layer + loss kernels (bg_head.hip) against plain torch fp32 / float64 on the same inputs.

Tolerances: outputs of the 128-term dot products 2e-5 relative to the row's scale (fp32 summation order differs from the library GEMM);
gradients that are sums over B rows 1e-4 relative to their largest entry; float64 statistics 1e-5 relative.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _data(B, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    dev = "cuda:0"
    h = torch.nn.functional.elu(torch.randn(B, 128, generator=g)).to(dev)
    W = (torch.randn(12, 128, generator=g) * 0.1).to(dev)
    b = (torch.randn(12, generator=g) * 0.1).to(dev)
    return g, dev, h, W, b


@pytest.mark.parametrize("B", [64, 1000, 98304])
def test_actor_head_forward_and_critic_forward(B):
    from booster_gym_amd.utils.utils import actor_head_forward, critic_head_forward

    g, dev, h, W, b = _data(B)
    mu = actor_head_forward(h, W, b, torch.empty(B, 12, device=dev))
    ref = (h.double() @ W.double().t() + b.double())
    assert (mu.double() - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())
    w1, b1 = W[:1].contiguous(), b[:1].contiguous()
    v = critic_head_forward(h, w1, b1, torch.empty(B, device=dev))
    refv = (h.double() @ w1.double().t() + b1.double()).squeeze(-1)
    assert (v.double() - refv).abs().max() < 2e-5 * max(1.0, refv.abs().max().item())


@pytest.mark.parametrize("B", [1000, 98304])
def test_actor_head_loss_backward_matches_torch(B):
    """Same inputs through (a) the fused head and (b) torch ops + bg_ppo_loss (itself pinned by the reference fixture in test_gpu_ppo.py)."""
    from booster_gym_amd.utils.utils import actor_head_loss_backward, head_scratch, ppo_loss_fused

    g, dev, h, W, b = _data(B, 1)
    A = 12
    logstd = torch.full((A,), -2.0, device=dev) + 0.1 * torch.randn(A, generator=g).to(dev)
    old_logstd = torch.full((A,), -2.0, device=dev)
    mu_ref = h @ W.t() + b
    old_mu = mu_ref + 0.02 * torch.randn(B, A, generator=g).to(dev)
    actions = old_mu + 0.135 * torch.randn(B, A, generator=g).to(dev)
    old_logp = (-0.5 * ((actions - old_mu) / old_logstd.exp()) ** 2 - old_logstd - 0.9189385332046727).sum(-1)
    adv = torch.randn(B, generator=g).to(dev)
    adv_stats = torch.stack([adv.double().sum(), (adv.double() ** 2).sum(), torch.tensor(float(B), dtype=torch.float64, device=dev)])
    values, returns = torch.randn(B, generator=g).to(dev), torch.randn(B, generator=g).to(dev)
    # (b) unfused
    gmu, gval = torch.empty(B, A, device=dev), torch.empty(B, device=dev)
    gls_ref, st_ref = torch.zeros(A, dtype=torch.float64, device=dev), torch.zeros(5, dtype=torch.float64, device=dev)
    ppo_loss_fused(mu_ref, logstd, actions, old_mu, old_logstd, old_logp, adv, adv_stats, values, returns, 0.2, 1.0, -0.01, gmu, gval, gls_ref, st_ref)
    g_hidden_ref = (gmu.double() @ W.double()) * torch.where(h > 0, torch.ones_like(h), h + 1).double()
    dW_ref, db_ref, dbh_ref = gmu.double().t() @ h.double(), gmu.double().sum(0), g_hidden_ref.sum(0)
    # (a) fused
    g_hidden = torch.empty(B, 128, device=dev)
    dW, db, dbh = torch.empty(A, 128, device=dev), torch.empty(A, device=dev), torch.empty(128, device=dev)
    gls, st = torch.zeros(A, dtype=torch.float64, device=dev), torch.zeros(5, dtype=torch.float64, device=dev)
    mu = torch.empty(B, A, device=dev)
    actor_head_loss_backward(h, W, b, logstd, actions, old_mu, old_logstd, old_logp, adv, adv_stats, 0.2, 1.0, -0.01, g_hidden, dW, db, dbh, gls, st,
                             head_scratch(dev), mu_out=mu)
    torch.cuda.synchronize()
    rel = lambda x, r: ((x.double() - r.double()).abs().max() / max(1e-30, r.double().abs().max())).item()
    assert rel(mu, mu_ref) < 2e-5
    # mu differs from the library GEMM in the last bits; the ratio's exp() amplifies that by |adv| / sigma^2 ~ 50
    assert rel(g_hidden, g_hidden_ref) < 2e-3
    assert rel(dW, dW_ref) < 2e-3 and rel(db, db_ref) < 2e-3 and rel(dbh, dbh_ref) < 2e-3
    assert rel(gls, gls_ref) < 2e-3
    assert rel(st[1:], st_ref[1:]) < 1e-4 and st[0] == 0


@pytest.mark.parametrize("B", [1000, 98304])
def test_critic_head_backward_matches_torch(B):
    from booster_gym_amd.utils.utils import critic_head_backward, head_scratch

    g, dev, h, W, b = _data(B, 2)
    w = W[:1].contiguous()
    values, returns = torch.randn(B, generator=g).to(dev), torch.randn(B, generator=g).to(dev)
    gval = 2.0 * (values.double() - returns.double()) / B
    g_ref = gval[:, None] * w.double() * torch.where(h > 0, torch.ones_like(h), h + 1).double()
    dw_ref, db_ref, dbh_ref = (gval[:, None] * h.double()).sum(0, keepdim=True), gval.sum().reshape(1), g_ref.sum(0)
    g_hidden = torch.empty(B, 128, device=dev)
    dw, db, dbh = torch.empty(1, 128, device=dev), torch.empty(1, device=dev), torch.empty(128, device=dev)
    st = torch.zeros(5, dtype=torch.float64, device=dev)
    critic_head_backward(h, w, values, returns, g_hidden, dw, db, dbh, st, head_scratch(dev))
    torch.cuda.synchronize()
    rel = lambda x, r: ((x.double() - r.double()).abs().max() / max(1e-30, r.double().abs().max())).item()
    assert rel(g_hidden, g_ref) < 1e-5 and rel(dw, dw_ref) < 1e-4 and rel(dbh, dbh_ref) < 1e-4
    assert abs(db.item() - db_ref.item()) < 1e-4 * max(gval.abs().sum().item() / B, 1e-6) * B ** 0.5 + 1e-7
    assert abs(st[0].item() - ((values.double() - returns.double()) ** 2).sum().item()) < 1e-5 * st[0].item()
    assert (st[1:] == 0).all()


def test_head_results_are_run_to_run_deterministic():
    from booster_gym_amd.utils.utils import critic_head_backward, head_scratch

    B = 98304
    g, dev, h, W, b = _data(B, 3)
    w = W[:1].contiguous()
    values, returns = torch.randn(B, generator=g).to(dev), torch.randn(B, generator=g).to(dev)
    outs = []
    for _ in range(2):
        g_hidden, dw, db, dbh = torch.empty(B, 128, device=dev), torch.empty(1, 128, device=dev), torch.empty(1, device=dev), torch.empty(128, device=dev)
        critic_head_backward(h, w, values, returns, g_hidden, dw, db, dbh, torch.zeros(5, dtype=torch.float64, device=dev), head_scratch(dev))
        outs.append((g_hidden.clone(), dw.clone(), db.clone(), dbh.clone()))
    for a, b2 in zip(*outs):
        assert torch.equal(a, b2)


def test_deferred_reductions_equal_the_immediate_finishes():
    """bg_actor_head_partial / bg_critic_head_backward_partial / bg_mlp_layer_backward_partial + ONE bg_reduce_group launch against the
    immediate forms (main kernel + its own finish launch): the heads' sums are added in the same fixed order -> bit-equal, float64 statistics
    and grad_logstd equal; the backward layer's column sums are added in another (fixed) order -> 1e-5 of the largest entry.  dL/dz outputs
    of the main kernels are untouched by the deferral."""
    from booster_gym_amd import _lib
    from booster_gym_amd.utils.utils import actor_head_loss_backward, critic_head_backward, head_scratch, reduce_group

    B, A = 98304, 12
    g, dev, h, W, b = _data(B, 4)
    logstd = torch.full((A,), -2.0, device=dev) + 0.1 * torch.randn(A, generator=g).to(dev)
    old_logstd = torch.full((A,), -2.0, device=dev)
    old_mu = h @ W.t() + b + 0.02 * torch.randn(B, A, generator=g).to(dev)
    actions = old_mu + 0.135 * torch.randn(B, A, generator=g).to(dev)
    old_logp = (-0.5 * ((actions - old_mu) / old_logstd.exp()) ** 2 - old_logstd - 0.9189385332046727).sum(-1)
    adv = torch.randn(B, generator=g).to(dev)
    adv_stats = torch.stack([adv.double().sum(), (adv.double() ** 2).sum(), torch.tensor(float(B), dtype=torch.float64, device=dev)])
    values, returns = torch.randn(B, generator=g).to(dev), torch.randn(B, generator=g).to(dev)
    w1 = W[:1].contiguous()
    # a backward layer 256 -> 128 (K = 256 columns of G, N = 128)
    G = torch.randn(B, 256, generator=g).to(dev); Wt = (torch.randn(128, 256, generator=g) * 0.05).to(dev); act = torch.nn.functional.elu(torch.randn(B, 128, generator=g)).to(dev)

    def run(defer):
        out = {}
        fa, fc, fl = (_lib.ReduceProblem(), _lib.ReduceProblem(), _lib.ReduceProblem()) if defer else (None, None, None)
        gh_a, dW, db, dbh = torch.empty(B, 128, device=dev), torch.full((A, 128), float("nan"), device=dev), torch.full((A,), float("nan"), device=dev), torch.full((128,), float("nan"), device=dev)
        gls, st = torch.zeros(A, dtype=torch.float64, device=dev), torch.zeros(5, dtype=torch.float64, device=dev)
        sa, sc = head_scratch(dev), head_scratch(dev)
        actor_head_loss_backward(h, W, b, logstd, actions, old_mu, old_logstd, old_logp, adv, adv_stats, 0.2, 1.0, -0.01, gh_a, dW, db, dbh, gls, st, sa, finish=fa)
        gh_c, dw, dbc, dbhc = torch.empty(B, 128, device=dev), torch.full((1, 128), float("nan"), device=dev), torch.full((1,), float("nan"), device=dev), torch.full((128,), float("nan"), device=dev)
        critic_head_backward(h, w1, values, returns, gh_c, dw, dbc, dbhc, st, sc, finish=fc)
        Gout, bg, scr = torch.empty(B, 128, device=dev), torch.full((128,), float("nan"), device=dev), torch.empty((B + 127) // 128 * 128, device=dev)
        lib = _lib.load()
        if defer:
            _lib.check(lib.bg_mlp_layer_backward_partial(B, 256, 128, _lib.ptr(G), _lib.ptr(Wt), _lib.ptr(act), _lib.ptr(Gout), _lib.ptr(bg), _lib.ptr(scr), fl,
                                                         _lib.current_stream_ptr()), "bg_mlp_layer_backward_partial")
            assert torch.isnan(dW).all() and torch.isnan(bg).all() and (st == 0).all()  # nothing reduced yet
            reduce_group([fa, fc, fl])
        else:
            _lib.check(lib.bg_mlp_layer_backward(B, 256, 128, _lib.ptr(G), _lib.ptr(Wt), _lib.ptr(act), _lib.ptr(Gout), _lib.ptr(bg), _lib.ptr(scr),
                                                 _lib.current_stream_ptr()), "bg_mlp_layer_backward")
        torch.cuda.synchronize()
        return dict(gh_a=gh_a, dW=dW, db=db, dbh=dbh, gls=gls, st=st, gh_c=gh_c, dw=dw, dbc=dbc, dbhc=dbhc, Gout=Gout, bg=bg)

    now, later = run(False), run(True)
    for k in ("gh_a", "dW", "db", "dbh", "gh_c", "dw", "dbc", "dbhc", "Gout"):
        assert torch.equal(now[k], later[k]), k
    assert torch.allclose(now["gls"], later["gls"], rtol=1e-12, atol=0) and torch.allclose(now["st"], later["st"], rtol=1e-12, atol=0)
    assert (now["bg"] - later["bg"]).abs().max() <= 1e-5 * now["bg"].abs().max() and torch.isfinite(later["bg"]).all()
    # argument errors
    assert _lib.load().bg_reduce_group(None, 1, None) == -1
    bad = _lib.ReduceProblem()
    assert _lib.load().bg_reduce_group((_lib.ReduceProblem * 1)(bad), 1, None) == -1


@pytest.mark.parametrize("T,N", [(24, 4096), (24, 100), (5, 17), (32, 16)])
def test_critic_values_and_gae_in_one_launch_equal_the_separate_launches(T, N):
    """bg_critic_values_gae against bg_critic_head_forward followed by bg_gae on the same inputs (time-outs, dones, ragged env counts): values,
    advantages, returns and the in-place time-out bootstrap of the rewards bit for bit; the float64 moments to 1e-12 (another fixed summation
    order); a second call (ticket left at zero, sums overwritten) gives the same bits; T > 32 is refused."""
    from booster_gym_amd.utils.utils import critic_head_forward, critic_values_gae, gae

    g = torch.Generator(device="cpu").manual_seed(7 + T + N)
    dev = "cuda:0"
    h = torch.nn.functional.elu(torch.randn((T + 1) * N, 128, generator=g)).to(dev)
    w, b = (torch.randn(1, 128, generator=g) * 0.1).to(dev), torch.randn(1, generator=g).to(dev)
    rew = torch.randn(T, N, generator=g).to(dev)
    dones = (torch.rand(T, N, generator=g) < 0.05).to(dev)
    touts = ((torch.rand(T, N, generator=g) < 0.03).to(dev)) & dones
    r0 = rew.clone()
    v0 = critic_head_forward(h, w, b, torch.empty((T + 1) * N, device=dev))
    adv0, ret0, s0 = gae(r0, dones, touts, v0[: T * N].view(T, N), v0[T * N :], 0.995, 0.95)
    scratch = torch.zeros(3 * ((N + 15) // 16) + 1, dtype=torch.float64, device=dev)
    for rep in range(2):
        r1 = rew.clone()
        v1, adv1, ret1 = torch.full(((T + 1) * N,), float("nan"), device=dev), torch.full((T, N), float("nan"), device=dev), torch.full((T, N), float("nan"), device=dev)
        s1 = torch.full((3,), 123.0, dtype=torch.float64, device=dev)  # must be overwritten, not added to
        critic_values_gae(h, w, b, r1, dones, touts, 0.995, 0.95, v1, adv1, ret1, s1, scratch)
        assert torch.equal(v1, v0) and torch.equal(adv1, adv0) and torch.equal(ret1, ret0) and torch.equal(r1, r0), rep
        assert torch.allclose(s1, s0, rtol=1e-12, atol=1e-9) and float(s1[2]) == T * N, (s1, s0)
        assert float(scratch[-1]) == 0.0
        if rep == 0:
            s_first = s1.clone()
        else:
            assert torch.equal(s1, s_first)
    # values given (the chained forward kernel's value head wrote them): only the GAE half runs, same results
    r2 = rew.clone()
    adv2, ret2 = torch.full((T, N), float("nan"), device=dev), torch.full((T, N), float("nan"), device=dev)
    s2 = torch.full((3,), 7.0, dtype=torch.float64, device=dev)
    vin = v0.clone()
    critic_values_gae(None, w, b, r2, dones, touts, 0.995, 0.95, vin, adv2, ret2, s2, scratch)
    # (the scan-only form runs 64 envs per workgroup instead of 16: the moments are added in another fixed order)
    assert torch.equal(vin, v0) and torch.equal(adv2, adv0) and torch.equal(ret2, ret0) and torch.equal(r2, r0)
    assert torch.allclose(s2, s_first, rtol=1e-12, atol=1e-9) and float(s2[2]) == T * N and float(scratch[-1]) == 0.0
    s3 = torch.full((3,), 9.0, dtype=torch.float64, device=dev)
    critic_values_gae(None, w, b, rew.clone(), dones, touts, 0.995, 0.95, v0.clone(), adv2, ret2, s3, scratch)
    assert torch.equal(s3, s2)  # deterministic
    if T == 32:
        big = torch.zeros(33, N, device=dev)
        with pytest.raises(RuntimeError, match="horizon"):
            critic_values_gae(torch.zeros(34 * N, 128, device=dev), w, b, big, big.bool(), big.bool(), 0.99, 0.95, torch.zeros(34 * N, device=dev), big.clone(),
                              big.clone(), s1, scratch)
