"""Parity of the fused PPO kernels (through the C ABI) with the reference-generated golden vectors (tests/golden/ppo_*.npz) and with
the torch restatement of the reference update loop (oracle/ppo_ref.py).  fp32; tolerances stated per assert."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
G = lambda name: np.load(os.path.join(HERE, "golden", name))
DEV = "cuda:0"


def test_gae_matches_reference_fixture():
    from booster_gym_amd.utils.utils import discount_values, gae

    d = G("ppo_gae.npz")
    t = lambda k: torch.tensor(d[k], device=DEV)
    adv = discount_values(t("rewards"), t("dones") | t("time_outs"), t("values"), t("last_values"), float(d["gamma"]), float(d["lam"]))
    assert torch.allclose(adv, t("advantages"), atol=2e-6)
    # fused form: time-out overwrite + returns + moments
    rew = t("rewards").clone()
    a2, ret, sums = gae(rew, t("dones"), t("time_outs"), t("values"), t("last_values"), float(d["gamma"]), float(d["lam"]))
    exp_rew = t("rewards").clone(); exp_rew[t("time_outs")] = t("values")[t("time_outs")]
    assert torch.equal(rew, exp_rew)
    from oracle.ppo_ref import discount_values as dv_ref

    a_ref = dv_ref(exp_rew.cpu(), (t("dones") | t("time_outs")).cpu(), t("values").cpu(), t("last_values").cpu(), float(d["gamma"]), float(d["lam"]))
    assert torch.allclose(a2.cpu(), a_ref, atol=2e-6) and torch.allclose(ret.cpu(), t("values").cpu() + a_ref, atol=2e-6)
    s = sums.cpu().numpy()
    assert abs(s[0] - a_ref.double().sum().item()) < 1e-3 and abs(s[1] - a_ref.double().square().sum().item()) < 1e-2 and s[2] == a_ref.numel()


def test_gae_edge_cases():
    from booster_gym_amd.utils.utils import gae

    # T = 1, every env done / timed out; N not a multiple of the block
    T, N = 1, 77
    rew, val, last = torch.rand(T, N, device=DEV), torch.randn(T, N, device=DEV), torch.randn(N, device=DEV)
    done = torch.ones(T, N, dtype=torch.bool, device=DEV); to = torch.zeros(T, N, dtype=torch.bool, device=DEV)
    adv, ret, sums = gae(rew.clone(), done, to, val, last, 0.99, 0.9)
    assert torch.allclose(adv, rew - val, atol=1e-6) and sums[2].item() == N


def _epoch_inputs():
    d = G("ppo_epoch.npz")
    t = lambda k: torch.tensor(d[k], device=DEV)
    from booster_gym_amd.utils.model import ActorCritic

    model = ActorCritic(12, 47, 14).to(DEV)
    model.load_state_dict({k[3:]: torch.tensor(d[k]) for k in d.files if k.startswith("sd_")})
    return d, t, model


def test_fused_loss_gradients_match_reference_fixture():
    """One mini-epoch (runner.py:132-163): losses, advantages and every parameter gradient vs the reference-computed fixture."""
    from booster_gym_amd.utils.utils import gae, gaussian_logp, ppo_loss_fused

    d, t, model = _epoch_inputs()
    T, N, A = 24, 64, 12
    B = T * N
    obs, priv, act = t("obses"), t("priv"), t("actions")
    values = model.critic(torch.cat((obs, priv), -1).reshape(B, -1)).squeeze(-1)
    last_values = model.critic(torch.cat((t("last_obs"), t("last_priv")), -1)).squeeze(-1).detach()
    rew = t("rewards").clone()
    adv, ret, sums = gae(rew, t("dones"), t("time_outs"), values.detach().view(T, N).contiguous(), last_values, 0.995, 0.95)
    assert torch.allclose(rew, t("rewards_after"), atol=1e-6)
    assert torch.allclose(adv, t("advantages"), atol=2e-5) and torch.allclose(ret, t("returns"), atol=2e-5)
    mu = model.actor(obs.reshape(B, -1))
    assert torch.allclose(mu, t("mu").reshape(B, A), atol=1e-5)
    logstd = model.logstd.detach().reshape(-1).contiguous()
    lp = gaussian_logp(mu.detach().contiguous(), logstd, act.reshape(B, A).contiguous())
    assert torch.allclose(lp, t("logp").reshape(B), atol=2e-4)
    g_mu, g_v = torch.zeros(B, A, device=DEV), torch.zeros(B, device=DEV)
    g_ls, stats = torch.zeros(A, dtype=torch.float64, device=DEV), torch.zeros(5, dtype=torch.float64, device=DEV)
    old_logstd = torch.tensor(d["old_logstd"], dtype=torch.float32, device=DEV).contiguous()
    ppo_loss_fused(mu.detach().contiguous(), logstd, act.reshape(B, A).contiguous(), t("old_mu").reshape(B, A).contiguous(), old_logstd,
                   t("old_logp").reshape(B).contiguous(), adv.view(B), sums, values.detach().contiguous(), ret.view(B), 0.2, 1.0, -0.01, g_mu, g_v, g_ls, stats)
    s = stats.cpu().numpy()
    losses = np.array([s[0] / B, s[1] / B, s[2] / (B * A), s[3] / B, s[4] / B])
    assert np.allclose(losses, d["losses"], rtol=2e-4, atol=1e-6), (losses, d["losses"])
    torch.autograd.backward([mu, values], [g_mu, g_v])
    for k, p in model.named_parameters():
        ref = t("grad_" + k)
        got = g_ls.float().view_as(ref) if k == "logstd" else p.grad
        assert torch.allclose(got, ref, rtol=2e-3, atol=2e-6 + 2e-4 * ref.abs().max().item()), k


def test_actor_sample_kernel_matches_torch_actor():
    from booster_gym_amd.utils.model import ActorCritic

    torch.manual_seed(0)
    model = ActorCritic(12, 47, 14).to(DEV)
    for n in (1, 37, 4096):
        obs = torch.randn(n, 47, device=DEV)
        mu, act = torch.empty(n, 12, device=DEV), torch.empty(n, 12, device=DEV)
        model.sample_actions(obs, act, seed=5, counter=3, mu_out=mu)
        with torch.no_grad():
            ref = model.actor(obs)
        assert torch.allclose(mu, ref, atol=2e-5), n
        z = (act - mu) / torch.exp(model.logstd.detach())
        if n == 4096:
            assert abs(z.mean().item()) < 0.02 and abs(z.std().item() - 1.0) < 0.02  # unit normal noise
        # same (seed, counter) -> same sample ; different counter -> different sample
        act2 = torch.empty_like(act); model.sample_actions(obs, act2, seed=5, counter=3)
        act3 = torch.empty_like(act); model.sample_actions(obs, act3, seed=5, counter=4)
        assert torch.equal(act, act2) and not torch.equal(act, act3)


def test_adam_step_matches_torch_adam_with_grad_clipping():
    from booster_gym_amd.utils.runner import FlatAdam

    torch.manual_seed(1)
    ps = [torch.nn.Parameter(torch.randn(300, 7, device=DEV)), torch.nn.Parameter(torch.randn(11, device=DEV))]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    fa = FlatAdam(ps, lr=3e-4)
    ref = torch.optim.Adam(qs, lr=3e-4)
    for it in range(5):
        gs = [torch.randn_like(p) * (3.0 if it % 2 == 0 else 0.01) for p in ps]  # exercises clip and no-clip
        fa.zero_grad()
        for p, q, g in zip(ps, qs, gs):
            p.grad.copy_(g); q.grad = g.clone()
        torch.nn.utils.clip_grad_norm_(qs, 1.0)
        ref.step(); fa.step()
        for p, q in zip(ps, qs):
            assert torch.allclose(p, q, rtol=1e-5, atol=1e-6), it
    sd = fa.state_dict()
    assert set(sd) == {"state", "param_groups"} and sd["state"][0]["exp_avg"].shape == (300, 7)


def test_adapt_lr_rule():
    from booster_gym_amd.utils.runner import FlatAdam

    fa = FlatAdam([torch.nn.Parameter(torch.zeros(4, device=DEV))], lr=1e-3)
    kl = torch.zeros(1, dtype=torch.float64, device=DEV)
    for kl_mean, expect in ((0.05, 1e-3 / 1.5), (0.001, 1e-3), (0.01, 1e-3)):  # > 2*desired: /1.5 ; < desired/2: *1.5 ; else unchanged
        kl.fill_(kl_mean * 100)
        fa.adapt_lr(kl, 100, 0.01)
        assert abs(fa.lr.item() - expect) < 1e-9, (kl_mean, fa.lr.item())
    fa.lr.fill_(1.2e-5); kl.fill_(1.0 * 100); fa.adapt_lr(kl, 100, 0.01)
    assert abs(fa.lr.item() - 1e-5) < 1e-12  # floor
    fa.lr.fill_(9e-3); kl.fill_(0.0); fa.adapt_lr(kl, 100, 0.01)
    assert abs(fa.lr.item() - 1e-2) < 1e-9  # ceiling


def test_optimizer_step_keeps_the_weight_mirrors_current():
    """bg_param_mirror: the optimiser launch writes a zero-padded copy and a transposed copy of chosen weight matrices together with the parameters
    (what utils/model.py otherwise does with strided torch copies before every forward / backward); bad descriptors are refused."""
    import ctypes
    from booster_gym_amd import _lib
    from booster_gym_amd.utils.runner import FlatAdam

    torch.manual_seed(5)
    shapes = [(256, 47), (256,), (128, 256), (7,)]
    ps = [torch.nn.Parameter(torch.randn(*sh, device=DEV)) for sh in shapes]
    fa = FlatAdam(ps, lr=1e-2)
    pad = torch.full((256, 64), 7.0, device=DEV)  # columns 47.. are the caller's zero padding: must not be touched
    pad[:, 47:] = 0.0
    wt = torch.empty(256, 128, device=DEV)
    off = lambda p: (p.data_ptr() - fa.flat.data_ptr()) // 4
    ms = [_lib.ParamMirror(off(ps[0]), 256, 47, 0, 64, 0, _lib.ptr(pad)), _lib.ParamMirror(off(ps[2]), 128, 256, 1, 128, 0, _lib.ptr(wt))]
    arr = (_lib.ParamMirror * 2)(*ms)
    stats = torch.zeros(5, dtype=torch.float64, device=DEV); acc = torch.zeros_like(stats); last = torch.zeros_like(stats)
    for it in range(3):
        for p in ps:
            p.grad.copy_(torch.randn_like(p))
        before = ps[0].detach().clone()
        fa.step_fused(stats, acc, last, 4, 100.0, 0.01, mirrors=arr)
        assert not torch.equal(before, ps[0].detach())
        assert torch.equal(pad[:, :47], ps[0].detach()) and float(pad[:, 47:].abs().sum()) == 0.0, it
        assert torch.equal(wt, ps[2].detach().t()), it
    bad = (_lib.ParamMirror * 1)(_lib.ParamMirror(off(ps[2]), 128, 256, 1, 64, 0, _lib.ptr(wt)))  # ld < rows for a transposed mirror
    with pytest.raises(RuntimeError, match="mirror"):
        fa.step_fused(stats, acc, last, 4, 100.0, 0.01, mirrors=bad)
    fa.step_count -= 1
    bad = (_lib.ParamMirror * 1)(_lib.ParamMirror(fa.flat.numel() - 10, 128, 256, 0, 256, 0, _lib.ptr(wt)))  # runs past the flat buffer
    with pytest.raises(RuntimeError, match="mirror"):
        fa.step_fused(stats, acc, last, 4, 100.0, 0.01, mirrors=bad)


def test_fused_optimizer_step_equals_the_separate_launches():
    """bg_optimizer_step (clip + Adam + KL learning-rate rule + statistics bookkeeping in one launch) against torch Adam with clip_grad_norm_, the
    reference's learning-rate rule (runner.py:174-180) and plain sums; log-std gradient delivered as float64; deterministic bit for bit."""
    from booster_gym_amd.utils.runner import FlatAdam

    torch.manual_seed(3)
    shapes = [(256, 47), (256,), (1, 12), (128, 130), (7,)]

    def make():
        torch.manual_seed(3)
        ps = [torch.nn.Parameter(torch.randn(*sh, device=DEV)) for sh in shapes]
        return ps, FlatAdam(ps, lr=1e-3)

    ps, fa = make()
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ref = torch.optim.Adam(qs, lr=1e-3)
    ls_off = (ps[2].grad.data_ptr() - fa.grad.data_ptr()) // 4
    stats = torch.zeros(5, dtype=torch.float64, device=DEV); acc = torch.zeros_like(stats); last = torch.zeros_like(stats)
    gls = torch.zeros(12, dtype=torch.float64, device=DEV)
    lr_ref, acc_ref = 1e-3, torch.zeros(5, dtype=torch.float64)
    count, desired = 1000.0, 0.01
    for it, kl_mean in enumerate((0.05, 0.001, 0.01, 0.0, 0.3)):
        gs = [torch.randn_like(p) * (2.0 if it % 2 == 0 else 0.003) for p in ps]  # exercises clip and no-clip
        for p, q, g in zip(ps, qs, gs):
            p.grad.copy_(g); q.grad = g.clone()
        ps[2].grad.fill_(123.0)  # must be overwritten from the float64 log-std gradient
        gls.copy_(gs[2].reshape(-1).double())
        st = torch.tensor([1.5 + it, -2.0, 0.25, 3.0, kl_mean * count], dtype=torch.float64)
        stats.copy_(st)
        for g_ in ref.param_groups:
            g_["lr"] = lr_ref
        torch.nn.utils.clip_grad_norm_(qs, 1.0)
        ref.step()
        fa.step_fused(stats, acc, last, 4, count, desired, grad_logstd=gls, ls_off=ls_off)
        if kl_mean > desired * 2:
            lr_ref = max(1e-5, lr_ref / 1.5)
        elif kl_mean < desired / 2:
            lr_ref = min(1e-2, lr_ref * 1.5)
        acc_ref += st
        for p, q in zip(ps, qs):
            assert torch.allclose(p, q, rtol=2e-5, atol=2e-6), it
        assert abs(fa.lr.item() - lr_ref) < 1e-9 * max(1.0, lr_ref / 1e-5), (it, fa.lr.item(), lr_ref)
        assert torch.allclose(last.cpu(), st) and torch.allclose(acc.cpu(), acc_ref)
        assert float(stats.abs().sum()) == 0.0 and float(gls.abs().sum()) == 0.0  # zeroed for the next mini-epoch
    assert fa.step_count == 5
    # deterministic: the same sequence again gives the same bits
    ps2, fa2 = make()
    torch.manual_seed(99)
    g_a = [torch.randn_like(p) for p in ps2]
    outs = []
    for _ in range(2):
        ps3, fa3 = make()
        for p, g in zip(ps3, g_a):
            p.grad.copy_(g)
        fa3.step_fused(torch.zeros(5, dtype=torch.float64, device=DEV), torch.zeros(5, dtype=torch.float64, device=DEV),
                       torch.zeros(5, dtype=torch.float64, device=DEV), 4, 10.0, 0.01)
        outs.append(fa3.flat.clone())
    assert torch.equal(outs[0], outs[1])


def _assert_same_adam_steps(name, p, q, start):
    """Two fp32 evaluations of the same update.  Adam normalises every step to ~lr per parameter whatever the gradient's size, so a rounding-size
    difference in a gradient becomes a FRACTION OF A STEP in the parameter -- and the whole step (both signs) for the few elements whose gradient
    is itself rounding noise.  Bound: all but 0.5 % of the elements within 2 % of the distance the tensor's parameters moved (floor: fp32
    resolution of the weights), none further apart than twice that distance."""
    moved = (q - start).abs().max().item()
    d = (p - q).abs()
    off = (d > 0.02 * moved + 2e-6).float().mean().item()
    assert off <= 0.005 and d.max().item() <= 2.0 * moved + 2e-6, (name, off, d.max().item(), moved)


@pytest.mark.parametrize("n,E,T", [(128, 3, 24), (256, 5, 24), (4096, 20, 24), (128, 2, 40), (100, 2, 24)])
def test_full_update_matches_reference_loop(n, E, T):
    """Runner.update() (fused kernels, flat Adam, device-side LR) vs oracle/ppo_ref.ppo_update_reference (the reference loop op by op)
    from the same weights on the same rollout data: parameters after E mini-epochs agree, and so do the logged losses and the LR.
    (4096, 20, 24) is the bench's shape (BASELINE configs[1]: 800 / 768 slabs, 20 optimiser steps with the KL rule moving the learning rate);
    (128, 2, 40): a horizon beyond the fused GAE launch's 32 steps (bg_critic_head_forward + bg_gae instead); (100, 2, 24): a batch that is not
    whole 128-row slabs (no forward passes during the rollout, ragged last slab everywhere)."""
    from booster_gym_amd.utils.config import load_cfg
    from booster_gym_amd.utils.model import ActorCritic
    from booster_gym_amd.utils.runner import Runner
    from oracle.ppo_ref import ppo_update_reference

    cfg = load_cfg("T1", {"env.num_envs": n, "terrain.type": "plane", "runner.mini_epochs": E, "runner.horizon_length": T})
    r = Runner(cfg=cfg)
    assert r._fused_gae == (T <= 32) and r._rollout_forward == (n % 128 == 0)
    obs, infos = r.env.reset()
    r.buffer["obses"][0].copy_(obs); r.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
    r.rollout()
    ref_model = ActorCritic(12, 47, 14).to(DEV)
    ref_model.load_state_dict(r.model.state_dict())
    b = r.buffer
    rewards_ref = b["rewards"].clone()
    stats_ref, lr_ref = ppo_update_reference(ref_model, torch.optim.Adam(ref_model.parameters(), lr=1e-5), b["obses"][:T].clone(), b["privileged_obses"][:T].clone(),
                                             b["actions"].clone(), rewards_ref, b["dones"].clone(), b["time_outs"].clone(), b["obses"][T].clone(),
                                             b["privileged_obses"][T].clone(), mini_epochs=E, learning_rate=1e-5)
    p_start = {k: p.detach().clone() for k, p in r.model.named_parameters()}
    acc = r.update()
    summ = r._summarize(acc)
    for (k, p), (k2, q) in zip(r.model.named_parameters(), ref_model.named_parameters()):
        assert k == k2
        _assert_same_adam_steps(k, p, q, p_start[k])
        if E <= 5:  # (element-wise closeness holds for a handful of optimiser steps; beyond that the bound above on steps gone astray is the statement)
            assert torch.allclose(p, q, rtol=1e-3, atol=2e-6), (k, (p - q).abs().max().item())
    assert torch.allclose(b["rewards"], rewards_ref, atol=1e-5)  # in-place time-out overwrite, repeated every mini-epoch
    for k in ("value_loss", "actor_loss", "bound_loss", "entropy", "kl_mean"):
        assert abs(summ[k] - stats_ref[k]) <= 2e-4 * max(1.0, abs(stats_ref[k])), (k, summ[k], stats_ref[k])
    assert abs(summ["lr"] - lr_ref) < 1e-9


def test_forward_passes_run_during_the_rollout_change_no_bit():
    """Runner.rollout() evaluates the first mini-epoch's forward passes (critic hidden layers + values, actor hidden layers + old mu, old log-probs)
    on every step's rows while the simulator runs the next step (side stream, the update's own kernels on 128-row slabs).  Against the same runner with
    that switched off (the passes run at the start of update(), reference order runner.py:123-133): identical bits in the parameters, the Adam
    moments, the loss statistics and the old log-probabilities after two whole iterations (the second rollout reads the weight copies the optimiser
    launch wrote), on 256 envs (two slabs per step and network) and with the command curriculum and rough terrain on."""
    from booster_gym_amd.utils.config import load_cfg
    from booster_gym_amd.utils.runner import Runner

    # (the last two: a horizon that the group size does not divide, and a group larger than the horizon -- every row still gets its pass exactly once)
    for over, group in (({"terrain.type": "plane"}, None), ({"terrain.type": "trimesh", "commands.curriculum": True}, None),
                        ({"terrain.type": "plane", "runner.horizon_length": 5, "env.num_envs": 128}, 2),
                        ({"terrain.type": "plane", "runner.horizon_length": 5, "env.num_envs": 128}, 8)):
        res = []
        for ahead in (True, False):
            cfg = load_cfg("T1", dict({"env.num_envs": 256, "runner.mini_epochs": 3, "basic.seed": 7}, **over))
            r = Runner(cfg=cfg)
            assert r._rollout_forward, "the default path must be the overlapped one at this shape"
            r._rollout_forward = ahead
            if group is not None:
                r._rollout_group = group
            obs, infos = r.env.reset()
            r.buffer["obses"][0].copy_(obs); r.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
            for _ in range(2):
                stats = r.iteration().clone()
            torch.cuda.synchronize()
            res.append((r.optimizer.flat.clone(), r.optimizer.exp_avg_sq.clone(), stats, r._old_logp.clone(), r._old_mu.clone(), r._values_all.clone(), r.buffer["obses"].clone()))
            del r
        for a, b in zip(*res):
            assert torch.equal(a, b)


def test_training_is_reproducible_run_to_run():
    """Two runners built from the same configuration and seed give the same bits after three whole iterations -- parameters, Adam moments, loss
    statistics, the rollout buffers and the command-curriculum grid -- on rough terrain with the curriculum on (the one place where the env uses
    float atomics: equal increments, so their order does not matter).  What the long runs show at scale (profiles/r05_train_10000it_trimesh*.json: 100
    logged rows identical across boxes and builds) as a test: every reduction of the update is fixed-order, every random draw is counter-based."""
    from booster_gym_amd.utils.config import load_cfg
    from booster_gym_amd.utils.runner import Runner

    res = []
    for _ in range(2):
        cfg = load_cfg("T1", {"env.num_envs": 512, "runner.mini_epochs": 4, "basic.seed": 11, "terrain.type": "trimesh", "commands.curriculum": True})
        r = Runner(cfg=cfg)
        obs, infos = r.env.reset()
        r.buffer["obses"][0].copy_(obs); r.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
        for _ in range(3):
            stats = r.iteration().clone()
        torch.cuda.synchronize()
        res.append((r.optimizer.flat.clone(), r.optimizer.exp_avg.clone(), r.optimizer.exp_avg_sq.clone(), r.optimizer.lr.clone(), stats, r.buffer["obses"].clone(),
                    r.buffer["rewards"].clone(), r.buffer["actions"].clone(), r.env.curriculum_prob.clone(), r.env.get_field("root_states").clone()))
        del r
    for k, (a, b) in enumerate(zip(*res)):
        assert torch.equal(a, b), f"item {k} differs between two identical runs"


SWITCHES = {
    # name: (runner attributes, MLPTrainer class attributes) -- every branch of Runner.update() / MLPTrainer that a switch or a shape can select
    "default": ({}, {}),
    # (environment switches and the attributes they set: BG_ONE_LAUNCH_TAIL -> _one_launch_tail, BG_SPLIT_CHAIN_CUS -> _split_chain_cus,
    #  BG_ONE_STREAM -> _one_stream, BG_ROLLOUT_FORWARD -> _rollout_forward, BG_ROLLOUT_FORWARD_GROUP -> _rollout_group, BG_DEFER_FINISH -> _defer_finish / _defer_serial (own test below),
    #  BG_CHAIN_SPLIT / BG_CHAIN_SPLIT_BWD / BG_CHAIN_ALTERNATE / BG_WGRAD_SPLIT -> MLPTrainer.CHAIN_SPLIT / CHAIN_SPLIT_BWD / CHAIN_ALTERNATE / WGRAD_SPLIT; BG_OWN_RCCL = 0 and 1: tests/test_gpu_rccl.py; BG_FWD_CHAIN_CUS / BG_BWD_CHAIN_CUS: test_cu_shares_of_the_chains_change_no_bit below)
    "tail_as_three_launches": ({"_one_launch_tail": False}, {}),                  # reduce_group, weight gradients + finish, optimizer_step (what ranks of a job run)
    "separate_optimizer_tail": ({"_fused_opt": False}, {}),                      # bg_adam_step + bg_adapt_lr + torch adds (first step after a restore)
    "gae_as_three_launches": ({"_fused_gae": False}, {}),                        # bg_critic_head_forward + fill + bg_gae (horizons beyond 32 steps)
    "values_from_stored_activations": ({"_chain_values": False}, {}),            # bg_critic_values_gae with its own output layer
    "output_layers_as_library_gemms": ({"_fused_head": False}, {}),              # torch GEMMs + bg_ppo_loss (models of other widths)
    "hidden_layers_one_launch_each": ({}, {"CHAIN": False}),                     # bg_mlp_layer_forward x 3 per network
    "hidden_layers_as_library_gemms": ({}, {"FUSED": False}),                    # torch.addmm + elu_, torch.mm + bg_elu_backward_colsum, bmm weight gradients
    "weight_gradients_as_library_gemms": ({}, {"FUSED_WGRAD": False}),           # split-K bmm + sum
    "fp32_mfma_chains": ({}, {"CHAIN_SPLIT": False}),                            # BG_CHAIN_SPLIT=0: forward chain and backward layers on the fp32 matrix pipe (round 5's default)
    "split_forward_fp32_mfma_backward": ({}, {"CHAIN_SPLIT_BWD": False}),        # BG_CHAIN_SPLIT_BWD=0: the chained split forward, one fp32-MFMA launch per backward layer
    "fp32_mfma_weight_gradients": ({}, {"WGRAD_SPLIT": 0}),                      # BG_WGRAD_SPLIT=0: the grouped weight gradients on the fp32 matrix pipe (bg_wgrad.hip; the default until late round 6)
    "plain_accumulation": ({}, {"CHAIN_ALTERNATE": False}),                      # BG_CHAIN_ALTERNATE=0: no slab accumulates the negated sums
    "chain_one_workgroup_per_slab": ({"_split_chain_cus": False}, {}),            # BG_SPLIT_CHAIN_CUS=0: the two forward launches share the chip by slabs instead of by CUs
    "backward_chain_one_workgroup_per_slab": ({"_split_bwd_chain_cus": False}, {}),  # BG_SPLIT_BWD_CHAIN_CUS=0: ... and the two backward launches
    "two_launches_on_two_streams": ({"_one_stream": False}, {}),                 # BG_ONE_STREAM=0: critic and actor chains as separate launches on two streams (until mid round 6)
    "rollout_forward_off": ({"_rollout_forward": False}, {}),
    "rollout_forward_one_step_per_group": ({"_rollout_group": 1}, {}),
    "rollout_forward_five_steps_per_group": ({"_rollout_group": 5}, {}),
}


def _update_under(switch):
    from booster_gym_amd.utils.config import load_cfg
    from booster_gym_amd.utils.model import MLPTrainer
    from booster_gym_amd.utils.runner import Runner

    attrs, cls_attrs = SWITCHES[switch]
    saved = {k: getattr(MLPTrainer, k) for k in cls_attrs}
    try:
        for k, v in cls_attrs.items():
            setattr(MLPTrainer, k, v)
        cfg = load_cfg("T1", {"env.num_envs": 256, "terrain.type": "plane", "runner.mini_epochs": 3, "basic.seed": 11})
        r = Runner(cfg=cfg)
        for k, v in attrs.items():
            assert hasattr(r, k)
            setattr(r, k, v)
        obs, infos = r.env.reset()
        r.buffer["obses"][0].copy_(obs); r.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
        start = r.optimizer.flat.clone()
        acc = r.iteration().clone()
        torch.cuda.synchronize()
        first = (r.optimizer.flat.clone(), acc, r._summarize(acc))
        acc2 = r.iteration().clone()  # a second iteration: its rollout reads the weight copies the optimiser launch wrote
        torch.cuda.synchronize()
        return start, first, (r.optimizer.flat.clone(), acc2, r.buffer["actions"].clone())
    finally:
        for k, v in saved.items():
            setattr(MLPTrainer, k, v)


@pytest.fixture(scope="module")
def default_update():
    return _update_under("default")


@pytest.mark.parametrize("switch", [k for k in SWITCHES if k != "default"])
def test_update_through_every_switch_matches_the_default(switch, default_update):
    """Every branch of the update path that an attribute, a class switch or a shape can select (the library-GEMM fallbacks included) runs a whole
    PPO iteration from the same seed and lands where the default path lands: parameters within 2 % of the distance they moved (different kernels
    round differently, and Adam turns a rounding-size gradient difference into a fraction of a step), loss statistics to 1e-3, the same learning
    rate.  The rollout-forward variants run the same kernels on the same numbers: identical bits, also after a second iteration."""
    start, (p0, a0, s0), (q0, b0, act0) = default_update
    _, (p1, a1, s1), (q1, b1, act1) = _update_under(switch)
    if switch.startswith("rollout_forward") or switch in ("chain_one_workgroup_per_slab", "backward_chain_one_workgroup_per_slab", "two_launches_on_two_streams"):
        assert torch.equal(p1, p0) and torch.equal(a1, a0) and torch.equal(q1, q0) and torch.equal(b1, b0) and torch.equal(act1, act0)
        return
    _assert_same_adam_steps(switch, p1, p0, start)
    for k in ("value_loss", "actor_loss", "bound_loss", "entropy", "kl_mean"):
        assert abs(s1[k] - s0[k]) <= 1e-3 * max(1.0, abs(s0[k])), (k, s1[k], s0[k])
    assert abs(s1["lr"] - s0["lr"]) < 1e-9


def test_cu_shares_of_the_chains_change_no_bit(monkeypatch):
    """The CU shares of the two networks inside the chained launches (Runner._plan_chain_split; BG_FWD_CHAIN_CUS / BG_BWD_CHAIN_CUS fix them for A/B
    runs, BG_SPLIT_CHAIN_CUS / BG_SPLIT_BWD_CHAIN_CUS = 0 give every slab its own workgroup) decide which workgroup walks which slab, nothing else:
    at a size where the planner splits the chip (2,048 envs: 400 + 384 slabs on 256 CUs), the planner's shares, other shares and no shares give the same
    bits -- in the one-launch form and as two launches on two streams."""
    from booster_gym_amd.utils.config import load_cfg
    from booster_gym_amd.utils.runner import Runner

    def run(env, one_stream=True, split=True):
        for k in ("BG_FWD_CHAIN_CUS", "BG_BWD_CHAIN_CUS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = Runner(cfg=load_cfg("T1", {"env.num_envs": 2048, "terrain.type": "plane", "runner.mini_epochs": 2, "basic.seed": 5}))
        r._one_stream = one_stream
        r._split_chain_cus = r._split_bwd_chain_cus = split
        obs, infos = r.env.reset()
        r.buffer["obses"][0].copy_(obs); r.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
        acc = r.iteration().clone()
        torch.cuda.synchronize()
        shares = (r._critic_tr.chain_workgroups, r._actor_tr.chain_workgroups, r._critic_tr.chain_bwd_workgroups, r._actor_tr.chain_bwd_workgroups)
        return r.optimizer.flat.clone(), acc, shares

    p0, a0, s0 = run({})
    assert s0[0] > 0 and s0[0] + s0[1] == s0[2] + s0[3] and s0[0] % 8 == 0 and s0[2] % 8 == 0, s0  # the planner split the chip, in multiples of the XCDs
    for env, one_stream, split in (({"BG_FWD_CHAIN_CUS": "168,88", "BG_BWD_CHAIN_CUS": "176,80"}, True, True), ({"BG_FWD_CHAIN_CUS": "256,256"}, True, True),
                                   ({}, False, True), ({"BG_BWD_CHAIN_CUS": "96,160"}, False, True), ({}, True, False)):
        p1, a1, s1 = run(env, one_stream, split)
        assert torch.equal(p1, p0) and torch.equal(a1, a0), (env, one_stream, split, s1, (p1 - p0).abs().max().item())
        if "BG_BWD_CHAIN_CUS" in env:
            assert list(s1[2:]) == [int(v) for v in env["BG_BWD_CHAIN_CUS"].split(",")]
        if not split:
            assert s1 == (0, 0, 0, 0)


def test_update_with_deferred_reductions_equals_update_with_immediate_ones():
    """Runner.update() with the small reductions deferred to one launch in front of the weight gradients (the default), to one launch on the side
    stream beside them (BG_DEFER_FINISH=2) and with every finish inside its chain (=0), from identical weights and rollout data: same parameters
    after 3 mini-epochs up to the summation order of the hidden layers' bias gradients, same loss statistics."""
    from booster_gym_amd.utils.config import load_cfg
    from booster_gym_amd.utils.runner import Runner

    res = []
    for defer, serial in ((True, True), (True, False), (False, True)):
        cfg = load_cfg("T1", {"env.num_envs": 128, "terrain.type": "plane", "runner.mini_epochs": 3})
        r = Runner(cfg=cfg)
        r._defer_finish, r._defer_serial = defer, serial
        obs, infos = r.env.reset()
        r.buffer["obses"][0].copy_(obs); r.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
        r.rollout()
        acc = r.update().clone()
        torch.cuda.synchronize()
        res.append((r.optimizer.flat.clone(), acc, r._summarize(acc)))
        del r
    p0, a0, s0 = res[2]
    for p1, a1, s1 in res[:2]:
        assert torch.allclose(p1, p0, rtol=1e-5, atol=1e-7), (p1 - p0).abs().max().item()
        assert torch.allclose(a1, a0, rtol=1e-4)
        assert abs(s1["lr"] - s0["lr"]) < 1e-12
    assert torch.equal(res[0][0], res[1][0])  # the two deferred forms run the same launches: identical bits


@pytest.mark.parametrize("B,C,with_act", [(98304, 256, True), (98304, 128, True), (3000, 256, True), (1000, 128, True), (777, 12, False), (98304, 1, False),
                                          (130, 20, True)])
def test_elu_backward_colsum_matches_torch(B, C, with_act):
    from booster_gym_amd import _lib

    torch.manual_seed(B + C)
    z = torch.randn(B, C, device=DEV)
    act = torch.nn.functional.elu(z)
    g = torch.randn(B, C, device=DEV)
    exp = g * torch.where(z > 0, torch.ones_like(z), act + 1.0) if with_act else g.clone()
    got, col = g.clone(), torch.zeros(C, device=DEV)
    scratch = torch.empty(((B + 127) // 128) * C, device=DEV)
    _lib.check(_lib.load().bg_elu_backward_colsum(B, C, _lib.ptr(got), _lib.ptr(act) if with_act else None, _lib.ptr(col), _lib.ptr(scratch),
                                                  _lib.current_stream_ptr()))
    assert torch.allclose(got, exp, atol=1e-6)
    assert torch.allclose(col, exp.double().sum(0).float(), rtol=1e-4, atol=1e-3)
