"""Golden vectors for the PPO math, produced by importing the REFERENCE's utils/utils.py and utils/model.py (torch + numpy only,
no stub needed) in the build container:   python tests/golden/make_ppo_fixtures.py
Stores inputs and expected outputs only (.npz)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from utils.model import ActorCritic  # noqa: E402  (reference)
from utils.utils import discount_values, surrogate_loss  # noqa: E402  (reference)

torch.manual_seed(99)
T, N = 24, 64
rewards, values, last_values = torch.rand(T, N), torch.randn(T, N), torch.randn(N)
dones = torch.rand(T, N) < 0.03
time_outs = torch.rand(T, N) < 0.02
adv = discount_values(rewards, dones | time_outs, values, last_values, 0.995, 0.95)
old_lp, lp, a = torch.randn(T * N) * 0.1 - 10, torch.randn(T * N) * 0.3 - 10, torch.randn(T * N)
sl = surrogate_loss(old_lp, lp, a)
np.savez_compressed(os.path.join(HERE, "ppo_gae.npz"), rewards=rewards.numpy(), values=values.numpy(), last_values=last_values.numpy(),
                    dones=dones.numpy(), time_outs=time_outs.numpy(), gamma=0.995, lam=0.95, advantages=adv.numpy(), sl_old_logp=old_lp.numpy(),
                    sl_logp=lp.numpy(), sl_adv=a.numpy(), sl_value=sl.item())

# one scripted mini-epoch of runner.py:123-174 on the reference model (runner.py itself is not importable: imageio/wandb/isaacgym)
model = ActorCritic(12, 47, 14)
with torch.no_grad():
    model.actor[6].weight.mul_(3.0)  # push some means beyond +-1 so that the bound loss is exercised
sd0 = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
obses, priv = torch.randn(T, N, 47), torch.randn(T, N, 14)
last_obs, last_priv = torch.randn(N, 47), torch.randn(N, 14)
with torch.no_grad():
    actions = model.act(obses).sample()
# an "old" policy slightly different from the current one, so ratio != 1 and the clip has both branches
with torch.no_grad():
    old_dist = model.act(obses)
    old_logp = old_dist.log_prob(actions).sum(dim=-1)
    old_mu, old_scale = old_dist.loc.clone(), old_dist.scale[0, 0].clone()  # [12]
    for p in model.actor.parameters():
        p.add_(torch.randn_like(p) * 0.02)
    model.logstd.add_(0.05)
sd1 = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
rew = rewards.clone()
vals = model.est_value(obses, priv)
lastv = model.est_value(last_obs, last_priv)
with torch.no_grad():
    rew[time_outs] = vals[time_outs]
    advantages = discount_values(rew, dones | time_outs, vals, lastv, 0.995, 0.95)
    returns = vals + advantages
    adv_n = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
value_loss = F.mse_loss(vals, returns)
dist = model.act(obses)
logp = dist.log_prob(actions).sum(dim=-1)
actor_loss = surrogate_loss(old_logp, logp, adv_n)
bound_loss = torch.clip(dist.loc - 1.0, min=0.0).square().mean() + torch.clip(dist.loc + 1.0, max=0.0).square().mean()
entropy = dist.entropy().sum(dim=-1)
loss = value_loss + actor_loss + 1.0 * bound_loss + (-0.01) * entropy.mean()
loss.backward()
kl = torch.sum(torch.log(dist.scale / old_dist.scale) + 0.5 * (torch.square(old_dist.scale) + torch.square(dist.loc - old_dist.loc)) / torch.square(dist.scale) - 0.5, axis=-1)
out = {"rewards_after": rew.numpy(), "advantages": advantages.numpy(), "returns": returns.numpy(), "adv_norm": adv_n.numpy(),
       "values": vals.detach().numpy(), "last_values": lastv.detach().numpy(), "mu": dist.loc.detach().numpy(), "logp": logp.detach().numpy(),
       "losses": np.array([value_loss.item(), actor_loss.item(), bound_loss.item(), entropy.mean().item(), kl.mean().item()])}
for k, p in model.named_parameters():
    out["grad_" + k] = p.grad.numpy()
np.savez_compressed(os.path.join(HERE, "ppo_epoch.npz"), obses=obses.numpy(), priv=priv.numpy(), last_obs=last_obs.numpy(), last_priv=last_priv.numpy(),
                    actions=actions.numpy(), rewards=rewards.numpy(), dones=dones.numpy(), time_outs=time_outs.numpy(), old_mu=old_mu.numpy(),
                    old_logstd=np.log(old_scale.numpy()), old_logp=old_logp.numpy(), **{"sd_" + k: v for k, v in sd1.items()}, **out)
print("losses", out["losses"])
