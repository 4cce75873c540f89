"""Actor-critic network.

Same architecture and `state_dict` key names as reference `utils/model.py:5-36` (so checkpoints and the exported
TorchScript actor interoperate): `critic.{0,2,4,6}`, `actor.{0,2,4,6}`, `logstd`; actor 47-256-128-128-12,
critic (47+14)-256-256-128-1, ELU, state-independent log-std initialised to -2.
GEMMs run through PyTorch-ROCm (hipBLASLt / rocBLAS, fp32 MFMA); the rollout-time inference + sampling is one
fused HIP launch (`sample_actions` -> bg_actor_sample).
"""
import torch

from .. import _lib

ACTOR_HIDDEN = (256, 128, 128)
CRITIC_HIDDEN = (256, 256, 128)


def _mlp(n_in, hidden, n_out):
    layers, prev = [], n_in
    for h in hidden:
        layers += [torch.nn.Linear(prev, h), torch.nn.ELU()]
        prev = h
    layers.append(torch.nn.Linear(prev, n_out))
    return torch.nn.Sequential(*layers)


class ActorCritic(torch.nn.Module):
    def __init__(self, num_act, num_obs, num_privileged_obs):
        super().__init__()
        self.critic = _mlp(num_obs + num_privileged_obs, CRITIC_HIDDEN, 1)
        self.actor = _mlp(num_obs, ACTOR_HIDDEN, num_act)
        self.logstd = torch.nn.parameter.Parameter(torch.full((1, num_act), fill_value=-2.0), requires_grad=True)

    def act(self, obs):
        mean = self.actor(obs)
        return torch.distributions.Normal(mean, torch.exp(self.logstd).expand_as(mean))

    def est_value(self, obs, privileged_obs):
        return self.critic(torch.cat((obs, privileged_obs), dim=-1)).squeeze(-1)

    # ---- fused rollout inference (reference runner.py:109-111: dist = model.act(obs); act = dist.sample())
    def sample_actions(self, obs, actions_out, seed, counter, mu_out=None):
        if not obs.is_cuda:
            raise RuntimeError("sample_actions runs the fused HIP actor kernel and needs CUDA tensors")
        a = self.actor
        w = [a[0].weight, a[0].bias, a[2].weight, a[2].bias, a[4].weight, a[4].bias, a[6].weight, a[6].bias, self.logstd]
        for t in w + [obs, actions_out]:
            if not t.is_contiguous():
                raise RuntimeError("sample_actions needs contiguous tensors")
        _lib.check(_lib.load().bg_actor_sample(obs.shape[0], _lib.ptr(obs), *[_lib.ptr(t) for t in w], int(seed), int(counter), _lib.ptr(mu_out),
                                               _lib.ptr(actions_out), _lib.current_stream_ptr()), "bg_actor_sample")
        return actions_out
