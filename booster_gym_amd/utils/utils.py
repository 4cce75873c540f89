"""Host-side mirrors of reference `utils/utils.py` on top of the HIP kernels.

  apply_randomization   reference utils/utils.py:5-30   (set-up time randomisation; same arguments,
                        same ValueError behaviour; per-step noise is drawn inside the HIP env kernel)
  discount_values       reference utils/utils.py:33-44  -> bg_gae (one backward scan, HBM-bound)
  ppo_loss_fused        reference utils/runner.py:144-174 + utils/utils.py:47-52 -> bg_ppo_loss
There is no torch/CPU fallback: CUDA tensors are required and the native library must load.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib


def apply_randomization(tensor, params, return_noise=False):
    if params is None:
        return tensor
    dist, op = params["distribution"], params["operation"]
    if dist == "gaussian":
        mu, std = params["range"]  # (mean, std): SURVEY Q8
        noise = torch.randn_like(tensor) if isinstance(tensor, torch.Tensor) else np.random.randn()
        value = mu + std * noise
    elif dist == "uniform":
        lo, hi = params["range"]
        noise = torch.rand_like(tensor) if isinstance(tensor, torch.Tensor) else np.random.rand()
        value = lo + (hi - lo) * noise
    else:
        raise ValueError(f"Invalid randomization distribution: {dist}")
    if op == "additive":
        out = tensor + value
    elif op == "scaling":
        out = tensor * value
    else:
        raise ValueError(f"Invalid randomization operation: {op}")
    return (out, noise) if return_noise else out


def rand_spec(params):
    """yaml randomisation entry -> (mode, a, b) of `bg_rand` (include/booster_gym_amd.h)."""
    if params is None:
        return (0, 0.0, 0.0)
    dist, op = params["distribution"], params["operation"]
    if dist not in ("gaussian", "uniform"):
        raise ValueError(f"Invalid randomization distribution: {dist}")
    if op not in ("additive", "scaling"):
        raise ValueError(f"Invalid randomization operation: {op}")
    mode = (1 if dist == "gaussian" else 3) + (0 if op == "additive" else 1)
    return (mode, float(params["range"][0]), float(params["range"][1]))


def _need_cuda(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("booster_gym_amd kernels need CUDA (ROCm) tensors; there is no CPU path")
        if not t.is_contiguous():
            raise RuntimeError("booster_gym_amd kernels need contiguous tensors")


def gae(rewards, dones, time_outs, values, last_values, gamma, lam, advantages=None, returns=None, sums=None):
    """Fused timeout-overwrite + GAE + returns + moments.  rewards is modified in place (runner.py:135).

    Returns (advantages [T,N], returns [T,N], sums float64[3] = (sum, sum of squares, count))."""
    T, N = rewards.shape
    advantages = torch.empty_like(values) if advantages is None else advantages
    returns = torch.empty_like(values) if returns is None else returns
    if sums is None:
        sums = torch.zeros(3, dtype=torch.float64, device=rewards.device)
    else:
        sums.zero_()
    d8 = dones.view(torch.uint8) if dones.dtype == torch.bool else dones
    t8 = time_outs.view(torch.uint8) if time_outs.dtype == torch.bool else time_outs
    _need_cuda(rewards, d8, t8, values, last_values, advantages, returns, sums)
    lib = _lib.load()
    _lib.check(lib.bg_gae(T, N, _lib.ptr(rewards), _lib.ptr(d8), _lib.ptr(t8), _lib.ptr(values), _lib.ptr(last_values), gamma, lam,
                          _lib.ptr(advantages), _lib.ptr(returns), _lib.ptr(sums), _lib.current_stream_ptr()), "bg_gae")
    return advantages, returns, sums


def critic_values_gae(h, w, b, rewards, dones, time_outs, gamma, lam, values_all, advantages, returns, sums, scratch):
    """Output layer of the critic on all (T + 1) N rows of h, then `gae` on the result, in one launch (bg_critic_values_gae); h = None: values_all is
    given (the chained forward kernel's value head wrote it) and only the GAE half runs.  rewards [T, N] is
    modified in place as in `gae`; values_all [(T + 1) N], advantages / returns [T, N] and sums float64[3] are written; scratch: float64
    [3 * ceil(N / 16) + 1] whose last element the caller zeroed once."""
    T, N = rewards.shape
    d8 = dones.view(torch.uint8) if dones.dtype == torch.bool else dones
    t8 = time_outs.view(torch.uint8) if time_outs.dtype == torch.bool else time_outs
    _need_cuda(rewards, d8, t8, values_all, advantages, returns, sums, scratch, *([] if h is None else [h, w, b]))
    if (h is not None and h.shape[0] != (T + 1) * N) or values_all.numel() != (T + 1) * N or scratch.numel() < 3 * ((N + 15) // 16) + 1:
        raise ValueError("critic_values_gae: h / values_all must have (T + 1) N rows, scratch 3 ceil(N / 16) + 1 float64")
    _lib.check(_lib.load().bg_critic_values_gae(T, N, _lib.ptr(h), _lib.ptr(w), _lib.ptr(b), _lib.ptr(rewards), _lib.ptr(d8), _lib.ptr(t8), gamma, lam,
                                                _lib.ptr(values_all), _lib.ptr(advantages), _lib.ptr(returns), _lib.ptr(sums), _lib.ptr(scratch),
                                                _lib.current_stream_ptr()), "bg_critic_values_gae")
    return values_all


def discount_values(rewards, dones, values, last_values, gamma, lam):
    """Reference signature (utils/utils.py:33): `dones` already includes the time-outs; rewards are not touched."""
    zeros = torch.zeros_like(dones, dtype=torch.uint8)
    adv, _, _ = gae(rewards.clone(), dones.contiguous(), zeros, values.contiguous(), last_values.contiguous(), gamma, lam)
    return adv


def gaussian_logp(mu, logstd, actions, out=None):
    B, A = mu.shape
    out = torch.empty(B, dtype=torch.float32, device=mu.device) if out is None else out
    _need_cuda(mu, logstd, actions, out)
    _lib.check(_lib.load().bg_gaussian_logp(B, A, _lib.ptr(mu), _lib.ptr(logstd), _lib.ptr(actions), _lib.ptr(out), _lib.current_stream_ptr()),
               "bg_gaussian_logp")
    return out


def ppo_loss_fused(mu, logstd, actions, old_mu, old_logstd, old_logp, adv, adv_stats, values, returns, e_clip, bound_coef, entropy_coef,
                   grad_mu, grad_values, grad_logstd, stats):
    """One pass over B samples: loss statistics and d(loss)/d(mu, values, logstd).  Outputs are caller-owned:
    grad_logstd float64[A] and stats float64[5] are zeroed here and accumulated with atomics."""
    B, A = mu.shape
    _need_cuda(mu, logstd, actions, old_mu, old_logstd, old_logp, adv, adv_stats, values, returns, grad_mu, grad_values, grad_logstd, stats)
    grad_logstd.zero_()
    stats.zero_()
    _lib.check(_lib.load().bg_ppo_loss(B, A, _lib.ptr(mu), _lib.ptr(logstd), _lib.ptr(actions), _lib.ptr(old_mu), _lib.ptr(old_logstd),
                                       _lib.ptr(old_logp), _lib.ptr(adv), _lib.ptr(adv_stats), _lib.ptr(values), _lib.ptr(returns), e_clip,
                                       bound_coef, entropy_coef, _lib.ptr(grad_mu), _lib.ptr(grad_values), _lib.ptr(grad_logstd), _lib.ptr(stats),
                                       _lib.current_stream_ptr()), "bg_ppo_loss")


def head_scratch(device):
    """Workspace of one fused head call (per-workgroup partial sums); calls that may run concurrently need their own."""
    return torch.empty(_lib.HEAD_SCRATCH_FLOATS, dtype=torch.float32, device=device)


def critic_head_forward(h, weight, bias, values_out):
    """values = h @ weight.T + bias for the 128 -> 1 output layer (model.py:21), one pass over h [rows, 128]."""
    _need_cuda(h, weight, bias, values_out)
    _lib.check(_lib.load().bg_critic_head_forward(h.shape[0], _lib.ptr(h), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(values_out),
                                                  _lib.current_stream_ptr()), "bg_critic_head_forward")
    return values_out


def actor_head_forward(h, weight, bias, mu_out):
    """mu = h @ weight.T + bias for the 128 -> 12 output layer (model.py:13) with the arithmetic of the fused loss kernel."""
    _need_cuda(h, weight, bias, mu_out)
    z = None
    _lib.check(_lib.load().bg_actor_head(h.shape[0], 0, _lib.ptr(h), _lib.ptr(weight), _lib.ptr(bias), z, z, z, z, z, z, z, 0.0, 0.0, 0.0,
                                         _lib.ptr(mu_out), z, z, z, z, z, z, z, _lib.current_stream_ptr()), "bg_actor_head")
    return mu_out


def actor_head_loss_backward(h, weight, bias, logstd, actions, old_mu, old_logstd, old_logp, adv, adv_stats, e_clip, bound_coef, entropy_coef,
                             g_hidden, grad_weight, grad_bias, grad_bias_hidden, grad_logstd, stats, scratch, mu_out=None, finish=None):
    """Output layer + PPO actor loss (runner.py:145-174) + output-layer backward in one pass over h [B, 128].  grad_logstd float64[12] and
    stats float64[5] (entries 1..4) are ACCUMULATED with atomics: the caller zeroes them (the critic head adds entry 0 concurrently).
    finish (a _lib.ReduceProblem): only the main kernel runs; the sums over its workgroups (grad_weight, grad_bias, grad_bias_hidden, grad_logstd,
    stats) are left to a later bg_reduce_group call on the descriptor written into `finish`; g_hidden is complete either way."""
    _need_cuda(h, weight, bias, logstd, actions, old_mu, old_logstd, old_logp, adv, adv_stats, g_hidden, grad_weight, grad_bias, grad_bias_hidden,
               grad_logstd, stats, scratch)
    if finish is not None:
        _lib.check(_lib.load().bg_actor_head_partial(h.shape[0], _lib.ptr(h), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(logstd), _lib.ptr(actions),
                                                     _lib.ptr(old_mu), _lib.ptr(old_logstd), _lib.ptr(old_logp), _lib.ptr(adv), _lib.ptr(adv_stats), e_clip,
                                                     bound_coef, entropy_coef, _lib.ptr(mu_out), _lib.ptr(g_hidden), _lib.ptr(grad_weight), _lib.ptr(grad_bias),
                                                     _lib.ptr(grad_bias_hidden), _lib.ptr(grad_logstd), _lib.ptr(stats), _lib.ptr(scratch), finish,
                                                     _lib.current_stream_ptr()), "bg_actor_head_partial")
        return
    _lib.check(_lib.load().bg_actor_head(h.shape[0], 1, _lib.ptr(h), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(logstd), _lib.ptr(actions),
                                         _lib.ptr(old_mu), _lib.ptr(old_logstd), _lib.ptr(old_logp), _lib.ptr(adv), _lib.ptr(adv_stats), e_clip,
                                         bound_coef, entropy_coef, _lib.ptr(mu_out), _lib.ptr(g_hidden), _lib.ptr(grad_weight), _lib.ptr(grad_bias),
                                         _lib.ptr(grad_bias_hidden), _lib.ptr(grad_logstd), _lib.ptr(stats), _lib.ptr(scratch),
                                         _lib.current_stream_ptr()), "bg_actor_head")


def critic_head_backward(h, weight, values, returns, g_hidden, grad_weight, grad_bias, grad_bias_hidden, stats, scratch, finish=None):
    """Backward of mean((values - returns)^2) (runner.py:148) through the 128 -> 1 output layer; stats[0] += sum of squared errors.
    finish: as for actor_head_loss_backward."""
    _need_cuda(h, weight, values, returns, g_hidden, grad_weight, grad_bias, grad_bias_hidden, stats, scratch)
    if finish is not None:
        _lib.check(_lib.load().bg_critic_head_backward_partial(h.shape[0], _lib.ptr(h), _lib.ptr(weight), _lib.ptr(values), _lib.ptr(returns), _lib.ptr(g_hidden),
                                                               _lib.ptr(grad_weight), _lib.ptr(grad_bias), _lib.ptr(grad_bias_hidden), _lib.ptr(stats),
                                                               _lib.ptr(scratch), finish, _lib.current_stream_ptr()), "bg_critic_head_backward_partial")
        return
    _lib.check(_lib.load().bg_critic_head_backward(h.shape[0], _lib.ptr(h), _lib.ptr(weight), _lib.ptr(values), _lib.ptr(returns), _lib.ptr(g_hidden),
                                                   _lib.ptr(grad_weight), _lib.ptr(grad_bias), _lib.ptr(grad_bias_hidden), _lib.ptr(stats),
                                                   _lib.ptr(scratch), _lib.current_stream_ptr()), "bg_critic_head_backward")


def reduce_group(problems):
    """Run the deferred reductions of a list of _lib.ReduceProblem descriptors in one launch on the current stream (bg_reduce_group)."""
    for k in range(0, len(problems), 8):  # the launch takes up to 8 descriptors (networks with more hidden layers than the reference's: several launches)
        part = problems[k : k + 8]
        arr = (_lib.ReduceProblem * len(part))(*part)
        _lib.check(_lib.load().bg_reduce_group(arr, len(part), _lib.current_stream_ptr()), "bg_reduce_group")


def surrogate_loss(old_actions_log_prob, actions_log_prob, advantages, e_clip=0.2):
    """Reference utils/utils.py:47-52, kept for API parity (plain torch ops on the caller's device; the
    training loop uses ppo_loss_fused instead)."""
    ratio = torch.exp(actions_log_prob - old_actions_log_prob)
    return torch.max(-advantages * ratio, -advantages * torch.clamp(ratio, 1.0 - e_clip, 1.0 + e_clip)).mean()
